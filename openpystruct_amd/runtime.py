"""Two properties of the process around the HIP library that decide whether captured HIP graphs are fast AND correct on this stack
(both found in r04, profiles/r04_notes.md; neither is in the kernels):

1. CPU threads vs the container's CPU quota.  The framework sizes its CPU thread pool by the host's core count (128 on a 256-core
   node) while the container may run under a CFS bandwidth quota (cpu.max, e.g. 16 CPUs per 100 ms period).  One CPU-side tensor op
   over more than 32 768 elements inside a GPU loop (r03: `torch.arange(50 000)` per generated shard) wakes that pool; its workers
   spin after the parallel region, the quota of the period is gone within ~10 ms, and EVERY thread of the container -- the one inside
   hipGraphLaunch included -- is frozen until the period ends: the "25-55 ms graph replay" of r03 (cgroup cpu.stat nr_throttled counts
   them; with one CPU thread there are none).  `fit_cpu_threads()` caps the pool at the usable cores.

2. Captured hipMemsetAsync nodes.  With the HIP runtime's graph packet capture on (its default, HIP 7.0.51831 in the torch wheel), a
   memset node writes the right value on the first launch of an instantiated graph and garbage on every later one
   (scripts/graph_reduce_probe.py).  The framework's multi-block reductions zero their semaphores with such a memset: a captured
   column sum over >= 512 rows -- the bias gradient of every nn.Linear / attention projection the framework differentiates itself --
   is stale or garbage from the second replay on: the "NaNs in bias gradients at the first replay after an eager pass" of r03 (the
   eager pass only re-shuffled which garbage the semaphores saw).  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 cures it at no measurable cost
   (profiles/r04_notes.md); it must be in the environment before the process's first HIP call: entry points call `configure()` first
   thing (r05: importing the package no longer touches the environment).  `graph_memsets_replay_correctly()` measures what the process actually got; `train_surrogate` captures
   framework-differentiated steps only when it says yes.  The library's own launches never use memset nodes.
"""
from __future__ import annotations

import math
import os
from typing import Dict

PACKET_CAPTURE_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def set_graph_env_defaults() -> None:
    """setdefault of the packet-capture switch: takes effect when no HIP call has been made yet in this process.  A user's value stays."""
    os.environ.setdefault(PACKET_CAPTURE_ENV, "0")


_CONFIGURED: Dict[str, object] = {}


def configure(graph_env: bool = True, cpu_threads: bool = True, log=None) -> Dict[str, object]:
    """What an ENTRY POINT (bench.py, a training script, a test session's conftest) calls once, as early as it can -- importing the
    package no longer edits the host process's environment (r05; VERDICT r04 weak 7 / ADVICE).

      * graph_env: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (item 2 above) unless the user set it.  Only a process that has not made a HIP
        call yet honours it; when the framework has already initialised the GPU the returned record says `graph_env_too_late` and
        `log` (if given) is told -- `graph_memsets_replay_correctly()` then measures what the process actually got, and
        `train_surrogate` captures framework-differentiated steps only when that probe passes.
      * cpu_threads: `fit_cpu_threads()` (item 1).

    Returns {'packet_capture_env', 'graph_env_too_late', 'cpu_threads', 'hip_runtime'}; idempotent."""
    import sys
    too_late = False
    if graph_env:
        torch = sys.modules.get("torch")
        # (only the framework's lazy initialisation is detected: an earlier torch.cuda.is_available() / ctypes call into the runtime also fixes
        #  what the runtime read from the environment -- graph_memsets_replay_correctly() measures what the process actually got)
        too_late = bool(torch is not None and torch.cuda.is_initialized()) and os.environ.get(PACKET_CAPTURE_ENV) is None
        set_graph_env_defaults()
        if too_late and log is not None:
            log(f"{PACKET_CAPTURE_ENV}=0 was set after this process's first HIP call: captured memset nodes may replay wrong values; "
                "framework-differentiated steps are captured only if runtime.graph_memsets_replay_correctly() passes")
    threads = fit_cpu_threads() if cpu_threads else None
    # (r06, ADVICE r05: configure() itself makes NO HIP call -- it is documented as the call to make before the first one, and bench.py forks
    #  its CPU baseline's process pool right after it.  The record carries the framework's HIP version string only; the runtime's own number
    #  is `hip_runtime_version(query_runtime=True)`, for after the process has chosen its device.)
    _CONFIGURED.update(packet_capture_env=os.environ.get(PACKET_CAPTURE_ENV), graph_env_too_late=too_late, cpu_threads=threads,
                       hip_runtime=hip_runtime_version(query_runtime=False))
    return dict(_CONFIGURED)


def hip_runtime_version(query_runtime: bool = True) -> str:
    """The HIP runtime this process runs on, as the framework reports it (torch.version.hip) plus -- `query_runtime`, which INITIALISES the HIP
    runtime: only after the process has forked whatever it forks -- the driver-side runtime number of the loaded library
    (hipRuntimeGetVersion) when a GPU is visible: the bench line records it so that a change of wheel is visible next to a change of numbers."""
    import sys
    torch = sys.modules.get("torch")
    if torch is None:
        import torch
    v = str(getattr(torch.version, "hip", None))
    if not query_runtime:
        return v
    try:
        if torch.cuda.is_available():
            import ctypes
            # the runtime THIS process has loaded (the framework's bundled copy), by the path it is mapped from: opening
            # "libamdhip64.so" by name could pull a second HIP runtime (the system's) into the process
            path = None
            for line in open("/proc/self/maps"):
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
            if path:
                rt = ctypes.CDLL(path)
                n = ctypes.c_int(0)
                if rt.hipRuntimeGetVersion(ctypes.byref(n)) == 0:
                    v += f" (hipRuntimeGetVersion {n.value})"
    except Exception:
        pass
    return v


def usable_cores() -> int:
    """Host cores this process may actually use: min(os.cpu_count(), scheduler affinity, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(math.ceil(q / per))))
            break
        except Exception:
            continue
    return n


def fit_cpu_threads(reserve: int = 2) -> int:
    """Cap the framework's CPU intra-op pool at the cores the container may use (minus `reserve` for the launching thread and the
    runtime's helper threads).  Returns the thread count in force.  Entry points (bench.py, the dataset generator's CLI) call it; the
    library does not change global framework settings on import."""
    import torch
    want = max(1, usable_cores() - reserve)
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)
    return torch.get_num_threads()


def cpu_throttle_counters() -> Dict[str, int]:
    """cgroup v2 cpu.stat: {'nr_periods', 'nr_throttled', 'throttled_usec'} (empty when not readable)."""
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k in ("nr_periods", "nr_throttled", "throttled_usec"):
                out[k] = int(v)
    except Exception:
        pass
    return out


_MEMSET_OK: Dict[int, bool] = {}


def graph_memsets_replay_correctly(device=None) -> bool:
    """Capture the framework's multi-block column sum (which zeroes its semaphores with hipMemsetAsync) in a HIP graph and replay it
    three times on fresh data: True when every replay equals the eager result.  ~1 ms, once per process and device."""
    import torch
    dev = torch.device("cuda" if device is None else device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _MEMSET_OK:
        return _MEMSET_OK[idx]
    with torch.cuda.device(idx):
        g = torch.randn(2048, 64, device=dev, dtype=torch.bfloat16)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            g.sum(0)
            side.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
                out = g.sum(0)
        torch.cuda.current_stream(dev).wait_stream(side)
        ok = True
        for _ in range(3):
            g.copy_(torch.randn(2048, 64, device=dev))
            ref = g.float().sum(0)
            gr.replay()
            torch.cuda.synchronize(dev)
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            ok = ok and err < 5e-2
        del gr
    _MEMSET_OK[idx] = ok
    return ok
