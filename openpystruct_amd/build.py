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
           os.path.join(CSRC, "fused_loss.hip"), os.path.join(CSRC, "beam_solve_lane.hip"), os.path.join(CSRC, "fused_bn.hip"), os.path.join(CSRC, "input_prep.hip"),
           os.path.join(CSRC, "mlp_block.hip"), os.path.join(CSRC, "seq_block.hip"), os.path.join(CSRC, "seq_layer.hip"), os.path.join(CSRC, "mem_bench.hip"), os.path.join(CSRC, "case_draw.hip")]
HEADERS = [os.path.join(CSRC, "beam_math.hpp"), os.path.join(CSRC, "beam_io.hpp"), os.path.join(CSRC, "call_counter.hpp"), os.path.join(CSRC, "dropout_stream.hpp"), os.path.join(CSRC, "input_noise.hpp"), os.path.join(CSRC, "sizing_math.hpp"), os.path.join(CSRC, "frame_wave.hpp"), os.path.join(CSRC, "frame_tile.hpp"), os.path.join(ROOT, "include", "openpystruct_amd.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-fast-math", "-ffp-contract=off", "-Wall", "-Wno-unused-command-line-argument",
           "-o", LIB + ".tmp"] + SOURCES
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    cmd[1:1] = os.environ.get("OPS_AMD_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DOPS_AMD_TRACE (diagnostic build)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
