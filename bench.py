#!/usr/bin/env python3
"""bench.py -- beam FE solves/s of the batched HIP solve (BASELINE.json metric, config 2).

One "step" = one pass of the hot path over one batch: B = 10 000 straight beams x 100 elements
(assemble + constrain + factor + solve + recover), inputs resident in HBM.  The K timed steps are
captured into one HIP graph (launch-bound inner loop) and replayed once inside the timed region.

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the cases are independent, so each rank solves its own B-beam shard (weak scaling);
there is NO collective in the data path -- only the barrier / max-over-ranks timing reduction.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.  Algorithmic bytes per solve: 4 925 B.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

# HIP graph packet capture off before anything can touch the GPU (N > 1: the process group is created before the package is imported):
# with it on, captured memset nodes replay garbage (openpystruct_amd/runtime.py item 2; no measurable cost, profiles/r04_notes.md)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# N > 1 ranks -- or OPS_AMD_FORCE_DP=1 with the launcher's one-rank environment (r05): the process group, the barriers, the timing
# all-reduce and the data-parallel training branch run over RCCL on the single GPU a test box has
FORCE_DP = os.environ.get("OPS_AMD_FORCE_DP", "0") == "1"


def is_dp(world):
    return world > 1 or FORCE_DP


BYTES_PER_SOLVE = 4925          # I[100]*8 + Fy[101]*8 + fix[101] in; v, theta [101]*8, V, M [100]*8 out
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
N_ELEM = 100
SEED = 20250307


CHIP_WARM_MS = 40.0      # untimed load in front of every timed region: from idle the chip needs ~25 ms of work before launch times settle
                         # (power 300 -> 670 W; scripts/hot_series.py, scripts/sat_series.py, profiles/r04_notes.md 6: 10^4 beams 12.8 -> 11.5 us,
                         # 2^20 beams 1 150 -> 934 us per launch over the first 30 launches) -- a generator runs for seconds, not for 6 ms


OWN_LOAD_WARM_MS = 60.0  # ... and of THAT time the last part has to be the workload itself (r05, scripts/region_transient_probe.py,
                         # profiles/r05_notes.md 14): behind 40 ms of the copy kernel the beam kernel starts at 12.9 us per launch and settles at
                         # 11.8 only after ~15 ms of ITS OWN execution (a 20-launch region: 13.07 us behind the copy kernel alone, 11.79 behind 20 ms
                         # more of the beam kernel, 11.86 behind 40 ms of the beam kernel and no copy kernel; 10 ms of idle undo it) -- the chip's
                         # operating point follows the instruction mix.  A 500-launch region warmed itself (its prelude replays the region twice); a
                         # 20-launch one at the driver's flags did not, which was the whole difference between the two on one box (12.7 / 11.5 us).
                         # `copy_warm_only` keeps the old prelude on the line, `cold_chip` none at all.


_WARM_BUF = {}


def chip_warm(fn=None, stream=None, ms=CHIP_WARM_MS, fn_ms=0.0):
    """`ms` of untimed load on the chip (the library's own copy kernel, csrc/mem_bench.hip, over 256 MiB: gets the chip out of idle whatever
    comes next), then (optionally) `fn` -- which queues the work about to be timed -- at least twice and for `fn_ms` (the chip's operating
    point follows the instruction mix: OWN_LOAD_WARM_MS), then drain."""
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    dev = torch.cuda.current_device()
    if dev not in _WARM_BUF:
        _WARM_BUF[dev] = (torch.empty(256 << 20, dtype=torch.uint8, device="cuda"), torch.empty(256 << 20, dtype=torch.uint8, device="cuda"))
    src, dst = _WARM_BUF[dev]
    s = stream if stream is not None else torch.cuda.current_stream()
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            lib.ops_hbm_copy16(src.data_ptr(), dst.data_ptr(), src.numel(), 0, s.cuda_stream)
        n += 8
        s.synchronize()
    if fn is not None:      # then the work itself: twice, or for `fn_ms`; short regions are queued several at a time (no idle between them)
        t1, k, reps = time.perf_counter(), 0, 1
        while k < 2 or (time.perf_counter() - t1) * 1e3 < fn_ms:
            tb = time.perf_counter()
            for _ in range(reps):
                fn()
            k += reps
            s.synchronize()
            if (time.perf_counter() - tb) * 1e3 < 1.0 and reps < 64:
                reps *= 2
    return n, (time.perf_counter() - t0) * 1e3


def synth_inputs(B, rank, device, inertia):
    """fixed bridge, 1-4 point loads per beam, UDL -1000, seeded per rank."""
    rng = np.random.default_rng(SEED + rank)
    N = N_ELEM + 1
    rollers = (10, 30, 70, 85, 100)
    fix = np.zeros(N, dtype=np.uint8)
    fix[0] = 1
    fix[[r - 1 for r in rollers]] = 1
    avail = np.array([n for n in range(2, N) if n not in rollers])
    k = rng.integers(1, 5, size=B)
    Fy = np.zeros((B, N))
    # vectorised draw of k distinct loaded nodes per beam
    order = np.argsort(rng.random((B, avail.size)), axis=1)[:, :4]
    vals = rng.uniform(-355857.0, -35585.7, size=(B, 4))
    for j in range(4):
        sel = k > j
        Fy[np.nonzero(sel)[0], avail[order[sel, j]] - 1] = vals[sel, j]
    if inertia == "uniform":
        I = np.full((B, N_ELEM), 0.5)
    elif inertia == "trajectory":
        I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=(B, N_ELEM)))
    else:
        I = np.exp(rng.uniform(np.log(1e-8), np.log(0.5), size=(B, N_ELEM)))
    t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device=device)  # noqa: E731
    return dict(x=t(np.linspace(0.0, 200.0, N)), E=t(200e9), I=t(I), fix=t(fix, torch.uint8), Fy=t(Fy), wy=t(-1000.0))


def cpu_baseline(B_hint):
    """Oracle (plain-C port, OpenMP over beams) on the host cores, bounded sample (~10-20 s)."""
    from oracle import beam_oracle as bo
    from oracle import c_oracle as co

    cores = usable_cores()
    gen = generator_baseline(cores)       # forks: before the OpenMP runtime of the FE leg has any threads
    rng = np.random.default_rng(SEED)
    x = np.linspace(0.0, 200.0, N_ELEM + 1)
    fix = bo.reference_fix_mask()
    nb = int(min(400000, max(20000, 2000 * cores)))          # >= 2000 beams per thread: amortises the parallel region
    I1, Fy1 = bo.random_cases(rng, 20000, inertia="trajectory")
    reps_tile = (nb + 19999) // 20000
    I, Fy = np.tile(I1, (reps_tile, 1))[:nb], np.tile(Fy1, (reps_tile, 1))[:nb]
    out = co.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF, n_threads=cores)  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    co.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF, n_threads=cores, out=out)
    dt = time.perf_counter() - t0
    reps = int(max(1, min(200, 10.0 / max(dt, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        co.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF, n_threads=cores, out=out)
    dt = time.perf_counter() - t0
    return {
        "value": nb * reps / dt,
        "unit": "beam FE solves/s",
        "cores": cores,
        "kind": "port",
        "generator": gen,
        "sample": f"{reps} x {nb} beams x {N_ELEM} elements, oracle/beam_oracle.c (band Cholesky, OpenMP static over beams, {cores} threads), "
                  f"{dt:.1f} s; OpenSeesPy itself unavailable (un-pinned third-party wheel)",
    }


def _gen_one(seed):
    """One sample of the reference's per-sample loop (MultiCore.py:130-240 `generate_sample`: sequential FE solve + torch-CPU
    autograd / Adam per epoch, patience 10) in a pool worker, one thread."""
    import torch as _t
    from oracle import beam_oracle as bo
    from oracle import sizing_oracle as so
    _t.set_num_threads(1)
    rs = np.random.default_rng(seed)
    x = np.linspace(0.0, 200.0, N_ELEM + 1)
    cand = [n for n in range(2, 101) if n not in bo.ROLLERS_REF]
    k = int(rs.integers(1, 5))
    rec = so.generate_sample(x, bo.ROLLERS_REF, rs.choice(cand, size=k, replace=False), rs.uniform(bo.MAX_FORCE, bo.MIN_FORCE, size=k),
                             patience=10, zero_last_node=True)
    return int(rec["epochs_run"])


def usable_cores():
    from openpystruct_amd import runtime
    return runtime.usable_cores()


def generator_baseline(cores):
    """CPU counterpart of MultiCore.py's main (MC:242-283): a process pool of `n_jobs` workers (the reference hard-codes 22,
    MC:52; here every host core), batches of samples handed out one per task, every sample the sequential per-epoch loop.
    Must run BEFORE this process touches the GPU (fork).  r06: ONE batch of the reference (500 samples, MC:246) or as many
    rounds of one-sample-per-core as 10 s hold, whichever ends first -- r05 measured 2 samples per worker in 0.85 s."""
    try:
        import multiprocessing as mp
        batch, budget_s = 500, 10.0
        ctx = mp.get_context("fork")
        t0 = time.perf_counter()
        ep, n = [], 0
        with ctx.Pool(processes=cores) as pool:
            pool.map(_gen_one, [SEED + 7], chunksize=1)                    # pool start-up + first-call costs stay outside
            t1 = time.perf_counter()
            while n < batch and (n == 0 or time.perf_counter() - t1 < budget_s):
                m = min(batch - n, max(cores, 8))                          # a round: one sample per worker
                ep += pool.map(_gen_one, [SEED + 100 + n + i for i in range(m)], chunksize=1)
                n += m
            dt = time.perf_counter() - t1
        return {"samples_per_s": n / dt, "samples_per_s_per_core": n / dt / cores, "samples": n, "seconds": dt, "n_jobs": cores,
                "pool_startup_s": t1 - t0, "mean_epochs_per_sample": float(np.mean(ep)), "reference_batch": batch,
                "what": "process pool of n_jobs workers, rounds of one sample per worker until one batch of the reference (500 samples, MC:246) or "
                        "10 s, per-sample sequential loop of the reference (MultiCore.py:242-283 structure; FE solve through the C port, torch CPU "
                        "autograd / Adam, 1 thread per worker)"}
    except Exception as e:   # the FE baseline must survive
        return {"error": repr(e)}


def stream_copy_gbs(dev, mib=1024, reps=10):
    """Achievable HBM rate on THIS box (report the fraction against the nominal peak and against a measured copy):
    device-to-device copy of `mib` MiB by the library's own 16-byte-per-lane copy kernel (csrc/mem_bench.hip; a framework
    `Tensor.copy_` reaches only 4.8-5.2 TB/s where the guide measures 6.29), read + write bytes over the HIP-event time; the
    better of plain and non-temporal stores."""
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    src = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    stream = torch.cuda.current_stream(dev).cuda_stream
    best = 0.0
    for nt in (0, 1):
        for _ in range(3):
            lib.ops_hbm_copy16(src.data_ptr(), dst.data_ptr(), src.numel(), nt, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.ops_hbm_copy16(src.data_ptr(), dst.data_ptr(), src.numel(), nt, stream)
        e1.record()
        torch.cuda.synchronize(dev)
        best = max(best, 2.0 * (mib << 20) * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def profiled_traffic(kernel_name, B, hint=""):
    """HBM bytes per launch from the newest committed PMC summary of this command (scripts/profile_gpu.sh:
    separate --pmc passes; FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE x2 for 16-byte coalesced reads on
    gfx950, MI355X_MICROARCH.md section HBM).  `hint`: substring of the profile's file name (hot / cold / sat) that
    tells runs of the same kernel and grid apart.  None when no matching profile is committed."""
    import glob
    import re
    best = None
    m = re.search(r"<(\d+),", kernel_name)
    bpw = 64 // int(m.group(1)) if m else 4
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            r = json.load(open(f))
        except Exception:
            continue
        if hint and hint not in os.path.basename(f):
            continue
        if kernel_name in r.get("kernel", "") and int(r.get("dispatch", {}).get("Grid_Size", 0)) == 64 * ((B + bpw - 1) // bpw):
            h = r.get("hbm", {})
            if h.get("FETCH_SIZE_raw") and h.get("WRITE_SIZE_raw"):
                best = (2.0 * h["FETCH_SIZE_raw"] + h["WRITE_SIZE_raw"]) * 1024.0, os.path.basename(f)
    return best


def profiled_frame_traffic(B, bays, stories):
    """HBM bytes per frame-solve launch from the committed PMC summary of the same command (profiles/*frames*pmc_summary.json), if any."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*frames*_pmc_summary.json"))):
        try:
            r = json.load(open(f))
        except Exception:
            continue
        if r.get("frames") == B and r.get("frame") == f"{bays}x{stories}":
            h = r.get("hbm", {})
            if h.get("FETCH_SIZE_raw") and h.get("WRITE_SIZE_raw"):
                best = (2.0 * h["FETCH_SIZE_raw"] + h["WRITE_SIZE_raw"]) * 1024.0, os.path.basename(f)
    return best


def epoch_median(ep):
    """Median of the epochs after the first (`epoch_s`): one epoch in a handful now and then takes ~9 ms longer on these boxes (a host hiccup, not the
    kernels: 144 epochs of `scripts/phys_bimodal_probe.py` show none) and a mean over two to four epochs carried it as +36 %
    (`tfd_physics` 0.0169 instead of 0.0124 s in three of ~15 r05 bench lines); the mean and every epoch stay on the line beside it."""
    s = sorted(ep)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


GRAD_BYTES = {"pinn": 593914 * 4, "tfd": 359876 * 4}          # ONE flat float32 all-reduce per step (2.38 MB / 1.44 MB)
XGMI_LINKS, XGMI_LINK_GBS, XGMI_EFF = 7, 153.0, 0.7             # the task's figures: 7 point-to-point links x ~153 GB/s per GPU; 70 % of a link assumed usable


def dp_projection(kind, step_plain_us, one_rank=None):
    """Projected data-parallel step and weak scaling at 2 / 4 / 8 ranks -- ON PAPER: no multi-GPU run exists for this code (one-GPU boxes only).
    Inputs: the plain step time measured in THIS run; the measured cost of the step's one-rank RCCL form over the plain step (this run's
    dp_segments if it ran data parallel, else profiles/r06_dp_one_rank_timing.json); the xGMI figures above.  Model of the exposed collective
    (S bytes, N ranks, per-step latency a):   ring  2 (N-1) (a + S / (N BW))   -- per-link bound, what a ring on point-to-point links costs;
    direct  2 (a + S / (N BW))  -- reduce-scatter + all-gather with every peer at once over its own link (what the full mesh allows).
    step_N = plain + max(one-rank overhead, collective); scaling_N = N plain / step_N.  `a` is NOT measured here: 2 and 5 us are shown."""
    S = GRAD_BYTES[kind]
    src = "this run (dp_segments)"
    if one_rank is None:
        try:
            r = json.load(open(os.path.join(ROOT, "profiles", "r06_dp_one_rank_timing.json")))
            one_rank = max(0.0, r[kind]["dp_one_graph_step_us"] - r[kind]["plain_step_us"])
            src = "profiles/r06_dp_one_rank_timing.json (one-graph step - plain step, one rank of RCCL)"
        except Exception:
            one_rank, src = 20.0, "assumed (no one-rank profile committed)"
    bw = XGMI_LINK_GBS * XGMI_EFF * 1e3           # bytes per us
    out = {"kind": kind, "grad_bytes": S, "step_plain_us": step_plain_us, "one_rank_overhead_us": one_rank, "one_rank_overhead_source": src,
           "assumptions": f"{XGMI_LINKS} xGMI links x {XGMI_LINK_GBS:.0f} GB/s per GPU at {XGMI_EFF:.0%}; per-step latency a in (2, 5) us: NOT measured; "
                          "step_N = plain + max(one-rank overhead, collective_N); nothing overlapped (the collective needs the last weight gradient)",
           "measured_multi_gpu": None, "ranks": {}}
    for N in (2, 4, 8):
        row = {}
        for a in (2.0, 5.0):
            ring = 2 * (N - 1) * (a + S / (N * bw))
            direct = 2 * (a + S / (N * bw))
            for name, c in (("ring", ring), ("direct", direct)):
                step = step_plain_us + max(one_rank, c)
                row[f"{name}_a{a:.0f}us"] = {"collective_us": round(c, 1), "step_us": round(step, 1), "scaling": round(N * step_plain_us / step, 2)}
        out["ranks"][str(N)] = row
    need = step_plain_us * (8.0 / 6.0 - 1.0)
    out["exposed_collective_for_6x_at_8_ranks_us"] = round(need, 1)
    return out


def surrogate_epoch_times(dev, rank, world, epochs, cases=50000):
    """Second half of BASELINE.json's metric: PINN / TFD epoch time (weak scaling: `cases` generated cases and the
    reference's batch size per GPU).  Returns a dict for the JSON line; never raises."""
    try:
        if is_dp(world):     # HIP events around the segments of the data-parallel step: the first multi-GPU run explains its own scaling
            os.environ.setdefault("OPS_AMD_DP_PROFILE", "1")
        from openpystruct_amd import dataprep, runtime, sizing, train
        thr0 = runtime.cpu_throttle_counters()
        t0 = time.perf_counter()
        rec = sizing.generate_dataset(cases * world, sizing.SizingConfig(), dev, rank=rank, world=world)
        torch.cuda.synchronize()
        cold = time.perf_counter() - t0           # first call: library load, first-use kernel loads, graph capture
        runs = []                                 # warm calls: the shard's state buffers and captured epoch graph are re-armed in place
        for _ in range(9):
            t0 = time.perf_counter()
            rec = sizing.generate_dataset(cases * world, sizing.SizingConfig(), dev, rank=rank, world=world)
            torch.cuda.synchronize()
            runs.append(time.perf_counter() - t0)
        out = {"generate_s": sorted(runs)[len(runs) // 2], "generate_s_median": sorted(runs)[len(runs) // 2], "generate_s_mean": sum(runs) / len(runs),
               "generate_s_max": max(runs), "generate_s_runs": runs, "generate_cold_s": cold, "cases_per_gpu": cases,
               "fe_solves_per_gpu": int(rec["epochs_run"].sum())}
        thr1 = runtime.cpu_throttle_counters()
        # the r03 "stalled replays" were CFS throttling of the container (runtime.py item 1): the counters of the generator leg
        out["runtime"] = {"cpu_threads": torch.get_num_threads(), "usable_cores": runtime.usable_cores(),
                          "cfs_throttled_periods_during_generate": (thr1.get("nr_throttled", 0) - thr0.get("nr_throttled", 0)) if thr0 and thr1 else None,
                          "graph_memsets_replay_correctly": runtime.graph_memsets_replay_correctly(dev),
                          "hip_graph_packet_capture_env": os.environ.get(runtime.PACKET_CAPTURE_ENV)}
        for kind in ("pinn", "tfd"):
            d = dataprep.prepare(rec, kind=kind, device=dev, distributed=is_dp(world))
            r = train.train_surrogate(kind, d, device=dev, max_epochs=epochs)
            ep = r["history"]["epoch_s"][1:] or r["history"]["epoch_s"]
            ep_s = sum(ep) / len(ep)             # r06: the metric is the MEAN epoch time, as in BENCH_r04 and before; the median beside it
            out[kind] = {"epoch_s": ep_s, "epoch_stat": "mean of the epochs after the first", "epoch_s_mean": ep_s, "epoch_s_median": epoch_median(ep),
                         "epoch_s_runs": ep, "steps_per_epoch": r["steps_per_epoch"],
                         "train_groups_per_gpu": int(d.X_train.shape[0]),
                         "dtype": "bf16", "step_us": 1e6 * ep_s / max(1, r["steps_per_epoch"]),
                         "step_us_median_epoch": 1e6 * epoch_median(ep) / max(1, r["steps_per_epoch"]),
                         # model quality of THIS short run (validation R^2 on un-standardised inertias, PINN:815-852 / TFD:800-829); trained to
                         # the reference's early stop on the same data the fast path reaches 0.66 (PINN) / 0.80 (TFD) like the framework path:
                         # profiles/r05_quality.json
                         "r2_val_I": float(r["r2_val_I"]), "epochs_run": int(r["epochs"]), "val_loss": float(r["history"]["val"][-1]),
                         "path": {"pinn": "layer-block launches (pinn_fused.py, csrc/mlp_block.hip)",
                                  "tfd": "one launch per encoder layer and direction + block launches (tfd_fused.py, csrc/seq_layer.hip, csrc/seq_block.hip)"}[kind]}
            if "dp_segments" in r:       # N > 1: device time of [graph A | all-reduce | graph B] per step (HIP events, mean over the epochs after the first)
                out[kind]["dp_segments"] = r["dp_segments"]
            if world == 1 and not is_dp(world):       # r06: what the measured one-rank figures project to at 2 / 4 / 8 ranks (a model, labelled as one)
                out[kind]["dp_projection"] = dp_projection(kind, out[kind]["step_us"])
        # BASELINE config 4: the TFD surrogate with the physics loss through the HIP FE-residual kernels.  The residual needs
        # per-case targets (n_cases = 1: 40 000 training rows per GPU instead of 6 666 groups), see DESIGN.md section 8
        scfg = sizing.SizingConfig()
        phys = train.PhysicsTerm(weight=1e-3, x=torch.linspace(0, scfg.L_max, scfg.num_nodes, dtype=torch.float64), E=scfg.E,
                                 fix=sizing.make_cases(1, scfg).fix[0], wy=scfg.uniform_udl)
        d1 = dataprep.prepare(rec, kind="tfd", n_cases=1, device=dev, distributed=is_dp(world))
        r = train.train_surrogate("tfd", d1, train.TfdConfig(n_cases=1), device=dev, max_epochs=max(2, epochs), physics=phys)
        ep = r["history"]["epoch_s"][1:] or r["history"]["epoch_s"]
        out["tfd_physics"] = {"epoch_s": sum(ep) / len(ep), "epoch_stat": "mean of the epochs after the first", "epoch_s_mean": sum(ep) / len(ep),
                              "epoch_s_median": epoch_median(ep), "epoch_s_runs": ep, "steps_per_epoch": r["steps_per_epoch"],
                              "train_rows_per_gpu": int(d1.X_train.shape[0]), "dtype": "bf16", "n_cases": 1}
        return out
    except Exception as e:   # the FE line must survive whatever happens here
        return {"error": repr(e)}


FRAME_BATCH = {"15x16": 12288, "10x10": 16384, "5x5": 32768, "3x3": 65536}       # frames per launch per GPU (the batches of profiles/*frames*_pmc_summary.json; 5 x 5: the
                                                                    # middle of the script's random range, FR:17-18; 3 x 3: the median half bandwidth of
                                                                    # its 100 (bays, stories) draws, 11 -- r06: 16 lanes per frame)


def frames_measure(dev, rank, local_rank, world, bays, stories, B, K, W):
    """BASELINE config 5: K launches of the batched frame solve over B frames of `bays` x `stories` per GPU (independent shards, no
    collective), HIP events on the launch stream, barriers outside them, max over ranks.  Every rank calls it; returns the record
    (the same on every rank)."""
    import torch.distributed as dist
    from openpystruct_amd import _cabi, frames

    topo = frames.grid_frame(bays, stories, device=dev)
    g = torch.Generator(device=dev).manual_seed(20250307 + rank)
    I = torch.exp(torch.empty((B, topo.Ne), dtype=torch.float64, device=dev).uniform_(math.log(1e-4), math.log(5e-3), generator=g))
    sol = frames.frame_solve(topo, I)
    for _ in range(W):
        frames.frame_solve(topo, I, out=sol)
    torch.cuda.synchronize()
    chip_warm(lambda: frames.frame_solve(topo, I, out=sol), fn_ms=OWN_LOAD_WARM_MS)      # untimed: the chip's power state settles (copy kernel, then the solve itself)

    def barrier():
        if is_dp(world):
            dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K):
        frames.frame_solve(topo, I, out=sol)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    dev_ms = e0.elapsed_time(e1)
    assert int(sol.status.abs().sum()) == 0
    if is_dp(world):
        tt = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(tt[0]), float(tt[1])
    # checker leg (rank 0, after the timed region, like cpu_baseline the only other place this file touches oracle/): 8 frames of the batch
    # the number is quoted on -- first, last, and six seeded picks -- against the oracle's dpbsv solve (the batch the
    # bench launches had only ever been checked for status == 0)
    checked, check_err, check_fail = 0, None, None
    if rank == 0:
        try:
            from oracle import beam_oracle as bo
            pick = sorted({0, B - 1} | {int(v) for v in np.random.default_rng(B).integers(0, B, size=6)})
            Ih, dh, fh = I[pick].cpu().numpy(), sol.disp[pick].cpu().numpy(), sol.forces[pick].cpu().numpy()
            check_err = 0.0
            for k in range(len(pick)):
                d, f, st, _, _ = bo.solve_model_3dof(topo.coords, topo.conn, topo.A, topo.E, Ih[k], topo.fix3, topo.nodal_loads, wy=topo.wy, wx=topo.wx)
                ed, ef = float(np.abs(dh[k] - d).max() / np.abs(d).max()), float(np.abs(fh[k] - f).max() / np.abs(f).max())
                check_err = max(check_err, ed, ef)
                if st != 0 or not (ed < 1e-7 and ef < 1e-6):
                    check_fail = f"frame {pick[k]}: oracle status {st}, displacement error {ed:.2e}, force error {ef:.2e}"
                    break
                checked += 1
        except Exception as e:          # reported in the record: an assertion here would leave the other ranks in the barrier below
            check_fail = repr(e)
    lib = _cabi.load()
    # the reference's own use of this solve is ONE frame per epoch (FR:178-183): time of a call on a batch of one (its own dispatch: the
    # workgroup-per-frame or four-waves-per-frame kernels), ten calls in a HIP graph, best of three replays.  Not part of `value`.
    one_us, one_family = None, None
    try:
        I1 = I[:1].contiguous()
        s1 = frames.frame_solve(topo, I1)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            frames.frame_solve(topo, I1, out=s1)
            side.synchronize()
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, stream=side):
                for _ in range(10):
                    frames.frame_solve(topo, I1, out=s1)
        torch.cuda.current_stream(dev).wait_stream(side)
        g1.replay(); torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        one_us = 1e9
        for _ in range(3):
            a0.record(); g1.replay(); a1.record(); torch.cuda.synchronize()
            one_us = min(one_us, a0.elapsed_time(a1) / 10 * 1e3)
        one_family = {0: "workgroup per frame (r01 kernels)", 3: "four waves per frame (csrc/frame_coop.hpp)"}.get(
            int(lib.ops_frame_plan_signature(1, topo.n_eq, topo.kd)) >> 24, "tuned")
    except Exception as e:              # the record must survive
        one_family = repr(e)
    # per frame, the call-wide assembly plan excluded (the packed kernel's share counts whole waves of 2 or 4 frames: difference over four frames)
    ws_frame = (int(lib.ops_frame_workspace_bytes(B + 4 - B % 4 + 4, topo.n_eq, topo.kd)) - int(lib.ops_frame_workspace_bytes(B + 4 - B % 4, topo.n_eq, topo.kd))) // 4
    # ALGORITHMIC bytes per frame: I in; disp [Nn,3], forces [Ne,6], V, M out -- what a solve that kept its factor on chip would move
    io_frame = 8 * (topo.Ne + 3 * topo.Nn + 8 * topo.Ne)
    us = dev_ms / K * 1e3
    achieved = io_frame * B / (us * 1e-6) / 1e9
    flops = 2.0 * topo.n_eq * topo.kd * topo.kd / 2.0            # band LDL^T multiply-adds (n kd^2 / 2), counted as 2 flop
    tr = profiled_frame_traffic(B, bays, stories)
    return {
        "metric": f"frame FE solves/s ({topo.Ne}-elem, batched)", "value": world * B * K / (dev_ms * 1e-3), "unit": "frame FE solves/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dev_ms / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic", "value_host_clock": world * B * K / wall,
        "solves_per_s": world * B * K / (dev_ms * 1e-3),
        "config": {"workload": f"BASELINE config 5: {B} frames of {bays}x{stories} bays x stories ({topo.Ne} elements, {topo.n_eq} "
                               f"equations, half bandwidth {topo.kd}) per GPU per step", "frames_per_step_per_gpu": B,
                   "workspace_bytes_per_frame": ws_frame, "workspace_bytes": ws_frame * B,
                   "parallelism": f"independent shards x{world}, no data-path collective"},
        # headline of this workload: the FP64 vector rate (the factorisation is n kd^2 flops on 42 KB of algorithmic I/O: it is
        # arithmetic-, not HBM-bound); the HBM record is on ALGORITHMIC bytes, with the measured (PMC) traffic named beside it
        "fp64_vector_frac": flops * B / (us * 1e-6) / 78.6e12,
        "batch_of_one": {"us_per_call": one_us, "kernel_family": one_family},
        "checked": checked, "checked_max_rel_err": check_err, "check_error": check_fail,     # frames of THIS batch compared with oracle.solve_model_3dof (disp 1e-7, forces 1e-6)
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": tr[0] if tr else None, "traffic_source": f"profiles/{tr[1]} (2*FETCH_SIZE + WRITE_SIZE, KiB)" if tr else None,
                     "traffic_over_algorithmic": tr[0] / (io_frame * B) if tr else None,
                     # the factor does not stay on chip (profiles/r06_notes.md 20): written once, read once = 2 * workspace_bytes_per_frame;
                     # the profiled traffic at THIS run's rate, as a fraction of the HBM peak (an upper reading: the Infinity Cache serves part of it)
                     "factor_round_trip_bytes_per_frame": 2 * ws_frame,
                     "traffic_rate_gbs": tr[0] / (us * 1e-6) / 1e9 if tr else None,
                     "traffic_rate_frac_of_hbm_peak": tr[0] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if tr else None,
                     "algorithmic_bytes_per_frame": io_frame, "kernel_us": us,
                     "fp64_vector_frac": flops * B / (us * 1e-6) / 78.6e12, "fp64_vector_peak_tflops": 78.6,
                     "flop_per_frame": flops},
    }


def bench_frames(args, rank, local_rank, world, dev):
    """`--workload frames`: the config-5 record as the line itself (the default line carries it under "frames")."""
    import torch.distributed as dist
    bays, stories = (int(v) for v in args.frame.split("x"))
    B = args.batch if args.batch != 10000 else FRAME_BATCH.get(args.frame, 12288)
    rec = frames_measure(dev, rank, local_rank, world, bays, stories, B, min(args.steps, 50), args.warmup)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if is_dp(world):
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=10000, help="beams per step per GPU (BASELINE config 2: 10000)")
    ap.add_argument("--workload", default="beams", choices=["beams", "frames"],
                    help="beams = BASELINE config 2 (the headline metric); frames = config 5 (batched 2-D frame solve)")
    ap.add_argument("--frame", default="15x16", help="bays x stories of the frames workload")
    ap.add_argument("--tiling", type=int, default=0, help="lanes per beam (0 = library default)")
    ap.add_argument("--inertia", default="trajectory", choices=["uniform", "trajectory", "adversarial"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sets", type=int, default=1, help="distinct input/output buffer sets the timed launches rotate over "
                                                        "(1 = BASELINE workload; 16 = the cache-defeating variant, for profiling)")
    ap.add_argument("--no-extras", action="store_true", help="skip the `cold` and `saturating` sub-records")
    ap.add_argument("--train-epochs", type=int, default=None,
                    help="also time N PINN / TFD training epochs on 50 000 generated cases per GPU (second half of the "
                         "BASELINE metric: data-parallel over the N ranks, one RCCL all-reduce per step); default: 5 at --gpus 1, "
                         "3 for N > 1; 0 switches it off")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if is_dp(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if FORCE_DP and world == 1:      # no launcher: the one-rank environment env:// needs
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        # nccl (= RCCL) in production; OPS_AMD_BENCH_BACKEND=gloo lets a 1-GPU box dry-run the N > 1 control path
        backend = os.environ.get("OPS_AMD_BENCH_BACKEND", "nccl")
        local_rank %= max(1, torch.cuda.device_count())
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)

    import openpystruct_amd as oa
    from openpystruct_amd import runtime

    # entry point: HIP graph environment default (also set at the top of this file, before the process group could touch the GPU) and the
    # framework's CPU pool inside the container's CPU quota (runtime.py items 1-2); the record goes into the line
    runtime_rec = runtime.configure(log=lambda m: print("warning: " + m, file=sys.stderr))

    # the CPU legs run first: the generator leg forks a process pool, which must happen before this process touches the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.workload == "beams":
        cpu = cpu_baseline(args.batch)

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    runtime_rec["hip_runtime"] = runtime.hip_runtime_version()      # (the runtime's own number: now that the CPU legs have forked and the device is chosen)
    if args.workload == "frames":
        return bench_frames(args, rank, local_rank, world, dev)
    B, K, W = args.batch, args.steps, args.warmup

    def barrier():
        if is_dp(world):
            dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()

    def measure(Bm, Km, Wm, n_sets, tiling, stream_out=False, warm_chip=True, own_warm=True):
        """K launches over `Bm` beams, rotating over `n_sets` distinct input / output buffer sets, captured in ONE HIP
        graph and replayed once between two HIP events on the launch stream.  Barriers and host synchronisation sit
        strictly OUTSIDE the event pair.  Returns (event ms, wall s, kernel name, launch mode)."""
        base = synth_inputs(Bm, rank, dev, args.inertia)
        sets = [base]
        for k in range(1, n_sets):      # distinct HBM: same geometry, rolled case order (values stay inside the distribution)
            sets.append(dict(base, I=base["I"].roll(k, 0).contiguous(), Fy=base["Fy"].roll(k, 0).contiguous()))
        outs = [oa.beam_solve(**st, tiling=tiling, stream_out=stream_out) for st in sets]       # allocates result buffers once per set
        torch.cuda.synchronize()
        stream = torch.cuda.Stream(device=dev)
        graph = None
        with torch.cuda.stream(stream):
            for i in range(Wm):
                oa.beam_solve(**sets[i % n_sets], tiling=tiling, out=outs[i % n_sets], stream_out=stream_out)
            stream.synchronize()
            if not args.no_graph:
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                        for i in range(Km):
                            oa.beam_solve(**sets[i % n_sets], tiling=tiling, out=outs[i % n_sets], stream_out=stream_out)
                    graph.replay()   # untimed: instantiate + first replay
                    stream.synchronize()
                    if warm_chip:
                        # untimed: CHIP_WARM_MS of the copy kernel, then OWN_LOAD_WARM_MS of the region itself (own_warm False: the region twice,
                        # the prelude of r04 / early r05; launches over GBs always took the longer form)
                        chip_warm(graph.replay, stream, fn_ms=OWN_LOAD_WARM_MS if own_warm else (CHIP_WARM_MS if Bm * BYTES_PER_SOLVE > (1 << 30) else 0.0))
                    else:            # `cold_chip`: exactly the driver's flags -- W warm-up launches, then let the chip fall back to idle
                        time.sleep(0.25)
                except Exception as e:   # keep the bench alive: eager launches measure the same kernel, with host gaps
                    print(f"warning: HIP graph capture failed ({e}); falling back to eager launches", file=sys.stderr)
                    graph = None
                    torch.cuda.synchronize()
        if graph is None and warm_chip:
            with torch.cuda.stream(stream):
                chip_warm(lambda: [oa.beam_solve(**sets[i % n_sets], tiling=tiling, out=outs[i % n_sets], stream_out=stream_out) for i in range(max(n_sets, 8))], stream)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            e0.record(stream)
            if graph is not None:
                graph.replay()
            else:
                for i in range(Km):
                    oa.beam_solve(**sets[i % n_sets], tiling=tiling, out=outs[i % n_sets], stream_out=stream_out)
            e1.record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0          # host clock around the same region, BEFORE the closing barrier
        barrier()
        dev_ms = e0.elapsed_time(e1)             # HIP events on the launch stream
        for o in outs:
            assert int(o.status.abs().sum()) == 0
        if is_dp(world):
            tt = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(tt[0]), float(tt[1])
        del outs, sets
        return dev_ms, wall, oa.kernel_name(Bm, N_ELEM, tiling), ("eager" if graph is None else f"one HIP graph of {Km} kernel nodes")

    def sub_record(Bm, Km, n_sets, tiling, what, stream_out=False, warm=None, warm_chip=True, own_warm=True):
        dev_ms, wall, kname, mode = measure(Bm, Km, min(W, 2 * n_sets) if warm is None else warm, n_sets, tiling, stream_out, warm_chip, own_warm)
        us = dev_ms / Km * 1e3
        ach = BYTES_PER_SOLVE * Bm / (us * 1e-6) / 1e9
        return {"what": what, "beams_per_launch_per_gpu": Bm, "launches": Km, "buffer_sets": n_sets,
                "distinct_bytes": n_sets * BYTES_PER_SOLVE * Bm, "kernel": kname, "kernel_us": us,
                "value": world * Bm * Km / (dev_ms * 1e-3), "achieved": ach, "frac": ach / HBM_PEAK_GBS, "launch": mode}

    import threading

    def fe_bail():       # a stalled collective of the FE part (barrier / time all-reduce) must not hang the driver: exit non-zero
        print(json.dumps({"metric": "beam FE solves/s (100-elem, batched)", "error": "a collective of the FE measurement stalled for 300 s",
                          "n_gpus": world}), flush=True)
        os._exit(3)

    fe_guard = threading.Timer(300.0, fe_bail)
    fe_guard.daemon = True
    if is_dp(world):
        fe_guard.start()
    dev_ms, wall, kname, mode = measure(B, K, W, max(1, args.sets), args.tiling)

    if rank == 0:
        kernel_ms = dev_ms / K
        achieved = BYTES_PER_SOLVE * B / (kernel_ms * 1e-3) / 1e9
        rec = {
            "metric": "beam FE solves/s (100-elem, batched)",
            # whole-job rate over the max-over-ranks HIP-event time of the K captured launches (barriers outside the
            # event pair: a 50-100 us barrier must not leak into a sub-millisecond timed region); host clock alongside
            "value": world * B * K / (dev_ms * 1e-3),
            "unit": "beam FE solves/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": kernel_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "value_host_clock": world * B * K / wall,
            "ms_per_step_host_clock": wall * 1e3 / K,
            "config": {
                "workload": f"BASELINE config 2: {B} beams x {N_ELEM} elements per GPU per step, fixed 5-roller bridge, "
                            f"1-4 point loads + UDL, inertia={args.inertia}, shared geometry (4925 B/solve); "
                            + ("ONE input/output buffer set re-solved every step: its 49 MB sit in the 256 MiB Infinity Cache, so `value` is a "
                               "cache-resident rate -- roofline.achieved / frac are the HBM-resident ones (16 rotating sets, 788 MB: the `cold` "
                               "record), roofline.frac_cache_resident this region's" if max(1, args.sets) == 1 and B * BYTES_PER_SOLVE < (200 << 20) else
                               f"{max(1, args.sets)} buffer set(s)"),
                "beams_per_step_per_gpu": B,
                "elements": N_ELEM,
                "kernel": kname,
                "launch": mode,
                "buffer_sets": max(1, args.sets),
                "parallelism": f"independent shards x{world}, no data-path collective",
                "untimed_chip_warm_ms": CHIP_WARM_MS,      # in front of EVERY timed region of this line (except `cold_chip`), beside the W warm-up steps
                "untimed_own_load_warm_ms": OWN_LOAD_WARM_MS,   # ... followed by this much of the region's own launches (except `cold_chip`, `copy_warm_only`)
                "hip_runtime": runtime_rec.get("hip_runtime"), "torch": torch.__version__,
                "hip_graph_packet_capture_env": runtime_rec.get("packet_capture_env"), "cpu_threads": runtime_rec.get("cpu_threads"),
            },
            # what the -m gpu parity tests hold this kernel to (tests/test_gpu_parity.py, tests/test_force_truth.py): FP64 throughout;
            # displacements <= 1e-10 (uniform I) / 1e-8 ("trajectory" I, this workload) relative to the oracle = north_star's 1e-6 with
            # margin; end forces V / M <= 2e-6 on the trajectory set -- the eps * kappa_s floor of ANY backward-stable solve of these
            # systems (the band solver the reference calls sits at 0.003-0.07 eps kappa_s, this kernel at 0.02-0.27; 50-digit truth)
            "parity": {"oracle": "oracle/beam_oracle.{py,c} (FE arithmetic unpinned: openseespy absent); generator loop pinned to the "
                                 "reference's own code (tests/golden/sizing_reference_*.npz)",
                       "displacements_rel": 1e-8, "end_forces_rel": 2e-6, "end_forces_bound": "eps * kappa_s (Jacobi-scaled condition), "
                       "tests/test_force_truth.py", "status_checked": True},
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel_us": kernel_ms * 1e3,
                "bytes_per_launch": BYTES_PER_SOLVE * B,
            },
        }
    else:
        rec = None
    # sub-records (every rank runs them: the timing reduction inside is collective)
    extras = {}
    if not args.no_extras and args.sets <= 1 and B == 10000:
        # cold: the same 10^4-beam launch over 16 distinct buffer sets (788 MB > the 256 MiB Infinity Cache): no read of the
        # timed region can be served by a cache that the previous replay filled -- the HBM claim without cache residency
        extras["cold"] = sub_record(B, max(32, min(K, 512) // 16 * 16), 16, args.tiling,
                                    "10^4-beam launches rotating over 16 distinct input/output sets (cache-defeating)")
        # the same with OPS_AMD_TILING_STREAM_OUT forced.  Since r03 the library picks the store policy itself from its estimate of
        # the result buffers' cache residency (csrc/beam_solve.hip, results_cache_resident): `cold` already runs non-temporal
        # stores and this record only confirms that the flag adds nothing
        extras["cold_stream_out"] = sub_record(B, max(32, min(K, 512) // 16 * 16), 16, args.tiling,
                                               "as `cold`, with the C ABI's streaming-output flag forced (non-temporal stores)", stream_out=True)
        # cold_chip: the headline launch with NO untimed load in front -- W warm-up launches, 0.25 s of idle, then the K timed launches: what a
        # 0.26 ms workload sees on a chip that was idle (the headline is the steady state of a chip under load)
        extras["cold_chip"] = sub_record(B, K, 1, args.tiling, f"the headline region without the {CHIP_WARM_MS:.0f} ms of untimed load: "
                                         f"{W} warm-up launches, 0.25 s idle, {K} timed launches", warm=W, warm_chip=False)
        # copy_warm_only: the headline region behind the prelude every bench line up to BENCH_r04 / profiles/r05_bench_driver_flags.json had
        # (40 ms of the copy kernel + the region twice): at the driver's K = 20 that is 0.5 ms of the beam kernel, and the region still sits in
        # the transient of its own instruction mix (OWN_LOAD_WARM_MS above)
        extras["copy_warm_only"] = sub_record(B, K, 1, args.tiling, f"the headline region behind {CHIP_WARM_MS:.0f} ms of the copy kernel and two "
                                              f"untimed replays of itself only (the prelude of r04 / early r05)", warm=W, own_warm=False)
        # saturating: SURVEY 8(d) asks for B = 2^20 next to the contract batch (one round of waves at 10^4 beams)
        # (12 untimed launches first: after the sub-millisecond launches above the first ~10 ms of 1 ms launches run 5-10 %
        #  slow -- clocks and TLBs of 5 GB of fresh buffers, measured with scripts/sat_ab.py)
        extras["saturating"] = sub_record(1 << 20, 10, 1, args.tiling, "2^20 beams per launch (5.2 GB per launch: HBM-resident by size)", warm=12)
    # BASELINE config 5 on the same line: the batched frame solve at the reference's largest frame (10 x 10), at the ~500-element one
    # BASELINE names (15 x 16), at the middle of the script's range (5 x 5) and at its median half bandwidth (3 x 3); <= 0.3 s each (20 launches
    # of 0.2-4 ms)
    frames_rec = {}
    if not args.no_extras and args.sets <= 1 and B == 10000:
        for fr in ("15x16", "10x10", "5x5", "3x3"):
            try:
                fb, fs = (int(v) for v in fr.split("x"))
                frames_rec[fr] = frames_measure(dev, rank, local_rank, world, fb, fs, FRAME_BATCH[fr], 20, 3)
            except Exception as e:       # the FE line must survive
                frames_rec[fr] = {"error": repr(e)}
                if world > 1:
                    break                # the ranks may have diverged inside the collective timing: no further frame collectives
    fe_guard.cancel()
    if rank == 0:
        rec.update(extras)
        if frames_rec:
            rec["frames"] = frames_rec
        # r06: the roofline object is the HBM one.  The headline region re-solves ONE 49 MB buffer set, which sits in the
        # 256 MiB Infinity Cache: its rate is kept as *_cache_resident; `achieved` / `frac` / `kernel_us` are the same launch over 16 rotating sets
        # (788 MB: no read can be served by what the previous launch left in a cache) -- the `cold` record.  `value` stays the headline region.
        rec["roofline"]["basis"] = "cache-resident (one 49 MB buffer set; the HBM-resident leg was not run)"
        if "cold" in extras:
            rl = rec["roofline"]
            rl["achieved_cache_resident"], rl["frac_cache_resident"], rl["kernel_us_cache_resident"] = rl["achieved"], rl["frac"], rl["kernel_us"]
            rl["achieved"], rl["frac"], rl["kernel_us"] = extras["cold"]["achieved"], extras["cold"]["frac"], extras["cold"]["kernel_us"]
            rl["frac_hbm_resident"], rl["kernel_us_hbm_resident"] = rl["frac"], rl["kernel_us"]        # (the r05 keys, same numbers)
            rl["basis"] = ("HBM-resident: 10^4-beam launches rotating over 16 distinct input/output sets (788 MB > the 256 MiB Infinity Cache), the "
                           "`cold` record; the headline region's own (Infinity-Cache-resident) rate is *_cache_resident")
        copy = stream_copy_gbs(dev)
        rec["roofline"]["stream_copy"] = copy
        rec["roofline"]["frac_of_stream_copy"] = rec["roofline"]["achieved"] / copy
        tr = profiled_traffic(rec["config"]["kernel"], B, "cold" if ("cold" in extras or args.sets > 1) else "hot") or profiled_traffic(rec["config"]["kernel"], B)
        if tr:
            rec["roofline"]["traffic"] = tr[0]
            rec["roofline"]["traffic_source"] = f"profiles/{tr[1]} (2*FETCH_SIZE + WRITE_SIZE, KiB)"
        for key, hint in (("cold", "cold"), ("cold_stream_out", "cold"), ("saturating", "sat")):
            if key in rec:
                tr = profiled_traffic(rec[key]["kernel"], rec[key]["beams_per_launch_per_gpu"], hint)
                rec[key]["traffic"] = tr[0] if tr else None
                if tr:
                    rec[key]["traffic_source"] = f"profiles/{tr[1]}"
        if cpu is not None:
            rec["cpu_baseline"] = cpu

    # second half of the metric, AFTER the FE record is complete.  Guard for N > 1: if a collective in the training
    # part ever stalls, rank 0 still prints the (already measured) FE line and every rank leaves.
    n_train = args.train_epochs if args.train_epochs is not None else (5 if world == 1 else 3)
    if n_train > 0:

        def bail():
            if rank == 0:
                rec["surrogate_epochs"] = {"error": "timed out after 420 s"}
                print(json.dumps(rec), flush=True)
            os._exit(3)          # a stalled collective is a failure: the FE line is out, the exit code says so

        guard = threading.Timer(420.0, bail)
        guard.daemon = True
        if is_dp(world):
            guard.start()
            os.environ.setdefault("OPS_AMD_DP_PROFILE", "1")      # (read by train.py at import: before the import below, as surrogate_epoch_times does)
            from openpystruct_amd import train as _train

            def stalled(what):           # the library's own stall timer (first replay of the one-graph step, exit 17): the FE line goes out first
                if rank == 0:
                    rec["surrogate_epochs"] = {"error": f"{what} stalled; OPS_AMD_DP_ONE_GRAPH=0 selects the two-graph step"}
                    print(json.dumps(rec), flush=True)
            _train.stall_hook = stalled
        info = surrogate_epoch_times(dev, rank, world, n_train)
        guard.cancel()
        if rank == 0:
            rec["surrogate_epochs"] = info
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if is_dp(world):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
