"""Builds the HIP shared library (C ABI, include/openpystruct_amd.h) in-tree with hipcc.

    python -m openpystruct_amd.build            # build if sources are newer than the .so
    python -m openpystruct_amd.build --force

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in CI containers too.
The .so is git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libopenpystruct_amd.so")
SOURCES = [os.path.join(CSRC, "beam_solve.hip"), os.path.join(CSRC, "beam_fat.hip"), os.path.join(CSRC, "sizing_step.hip"),
           os.path.join(CSRC, "beam_residual.hip"), os.path.join(CSRC, "frame_solve.hip"),
           os.path.join(CSRC, "stencil_bn.hip"), os.path.join(CSRC, "flat_adam.hip"),
           os.path.join(CSRC, "fused_loss.hip"), os.path.join(CSRC, "fused_bn.hip"), os.path.join(CSRC, "input_prep.hip"),
           os.path.join(CSRC, "mlp_block.hip"), os.path.join(CSRC, "seq_block.hip"), os.path.join(CSRC, "seq_layer.hip"), os.path.join(CSRC, "mem_bench.hip"), os.path.join(CSRC, "case_draw.hip")]
HEADERS = [os.path.join(CSRC, "repack_tiles.hpp"), os.path.join(CSRC, "beam_math.hpp"), os.path.join(CSRC, "beam_io.hpp"), os.path.join(CSRC, "call_counter.hpp"), os.path.join(CSRC, "dropout_stream.hpp"), os.path.join(CSRC, "input_noise.hpp"), os.path.join(CSRC, "sizing_math.hpp"), os.path.join(CSRC, "frame_wave.hpp"), os.path.join(CSRC, "frame_pack.hpp"), os.path.join(CSRC, "frame_coop.hpp"), os.path.join(ROOT, "include", "openpystruct_amd.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


_DEPS_CACHE = {}


def _deps(path: str, seen=None) -> set:
    """`path` and every file it #includes with quotes, transitively (the library's own headers; system headers do not change)."""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
        _deps(os.path.join(os.path.dirname(path), m.group(1)), seen)
    return seen


def _flags() -> list:
    return ([f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-Wall",
             "-Wno-unused-command-line-argument"] + os.environ.get("OPS_AMD_EXTRA_HIPCC_FLAGS", "").split())   # e.g. -DOPS_AMD_TRACE (diagnostic build)


def _obj(src: str) -> str:
    return os.path.join(os.path.dirname(LIB), "obj", os.path.splitext(os.path.basename(src))[0] + ".o")


def _stale_objects(force: bool) -> list:
    """Sources whose object is older than the source, one of its headers, this file, or was built with other flags."""
    stamp = os.path.join(os.path.dirname(LIB), "obj", "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(_flags())
    out = []
    for src in SOURCES:
        o = _obj(src)
        if force or not same_flags or not os.path.exists(o):
            out.append(src)
            continue
        t = os.path.getmtime(o)
        if src not in _DEPS_CACHE:                      # (the include scan once per process, not on every needs_build())
            _DEPS_CACHE[src] = _deps(src)
        if any(os.path.getmtime(f) > t for f in _DEPS_CACHE[src] | {os.path.abspath(__file__)}):
            out.append(src)
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB) or _stale_objects(False):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(_obj(s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """One object per source (only the stale ones, in parallel), then one link.  r05: the whole library in one hipcc call took ~6 min
    on the build container's 8 cores; a change to one kernel file now costs that file's compile + the link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.join(os.path.dirname(LIB), "obj"), exist_ok=True)
    todo = _stale_objects(force)
    hipcc, flags = hipcc_path(), _flags()
    if verbose:
        flags = ["-Rpass-analysis=kernel-resource-usage"] + flags

    def compile_one(src):
        subprocess.check_call([hipcc] + flags + ["-c", "-o", _obj(src) + ".tmp", src])
        os.replace(_obj(src) + ".tmp", _obj(src))

    jobs = jobs or int(os.environ.get("OPS_AMD_BUILD_JOBS", "0")) or min(len(todo) or 1, max(1, (os.cpu_count() or 2) - 1))
    # (largest files first: the long compiles should not start last)
    todo.sort(key=lambda f: -os.path.getsize(f))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    with open(os.path.join(os.path.dirname(LIB), "obj", "flags.txt"), "w") as f:
        f.write(" ".join(_flags()))
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp"] + [_obj(s) for s in SOURCES])
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
