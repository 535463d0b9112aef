"""The two process-level findings of r04 (openpystruct_amd/runtime.py, profiles/r04_notes.md) as regression tests: neither was in a
kernel, both decided whether captured HIP graphs were correct / fast."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_captured_memset_nodes_replay_correctly_in_this_process():
    """tests/conftest.py (an entry point: what `runtime.configure()` does) puts DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 into the environment
    before the first HIP call: the framework's multi-block reductions -- which zero their semaphores with hipMemsetAsync -- then give the eager result
    on every replay of a captured graph.  With the runtime's default they are right on the first replay only."""
    from openpystruct_amd import runtime
    assert os.environ.get(runtime.PACKET_CAPTURE_ENV) == "0"
    assert runtime.graph_memsets_replay_correctly("cuda")
    # the reduction that carried the r03 NaNs, checked directly: bias gradient of a bf16 Linear over 3584 rows, four replays
    lin = torch.nn.Linear(120, 360).cuda()
    x = torch.randn(3584, 120, device="cuda")

    def step():
        lin.weight.grad = lin.bias.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = lin(x)
        (y.float() ** 2).sum().backward()
        return lin.bias.grad

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = step()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(4):
        x.copy_(torch.randn_like(x))
        ref = step().clone()
        g.replay()
        torch.cuda.synchronize()
        assert float((out - ref).abs().max() / ref.abs().max()) < 5e-2


def test_framework_differentiated_tfd_runs_stay_finite_with_eager_steps_between_replays(monkeypatch):
    """The r03 anomaly: a framework-path Transformer-Diffusion run after a fast-path run in one process, explicit root gradient, the
    epoch's tail batch and validation pass run EAGERLY between graph replays -> NaNs in the attention's bias gradients in 6-9 of 12
    runs (gpurun_out -> profiles/r04_nan_hunt.log).  Cause: captured memset nodes (see above), not the step.  Five framework runs."""
    from openpystruct_amd import dataprep, sizing, tfd_fused, train
    from openpystruct_amd import switches
    monkeypatch.setitem(switches._values, "tail_graph", "0")
    assert train._EXPLICIT_ROOT
    rec = sizing.generate_dataset(6000, sizing.SizingConfig(max_e=60), "cuda")
    d = dataprep.prepare(rec, kind="tfd", device="cuda")
    for fast in (True, False, False, False, False, False):
        monkeypatch.setattr(tfd_fused, "ENABLED", fast)
        out = train.train_surrogate("tfd", d, device="cuda", max_epochs=6, seed=1)
        h = out["history"]
        assert np.all(np.isfinite(h["train"])) and np.all(np.isfinite(h["val"])), (fast, h)
        assert all(bool(torch.isfinite(q).all()) for q in out["model"].parameters())


def test_generator_shards_neither_stall_nor_get_the_container_throttled():
    """r03: every third warm 50 000-case shard took 60-70 ms instead of 22 -- one graph launch frozen for the rest of a 100 ms scheduler
    period.  Cause: `torch.arange(50 000)` on the CPU per shard woke the framework's 128-thread pool, whose spinning workers used up
    the container's CPU quota (cgroup cpu.max) within the period.  With the default thread count still in force: twelve warm shards,
    the cgroup's throttle counter does not move, and the shard times show no such pattern: at most ONE of the twelve above twice the
    median and none above four times (r05: on a busy node a single shard is occasionally 2-3 x slow -- 2 of 10 full-suite runs on one
    box, never when the file runs alone -- while the r03 pathology made every third shard 3 x slow)."""
    from openpystruct_amd import runtime, sizing
    cfg = sizing.SizingConfig()
    sizing.generate_dataset(50000, cfg, "cuda")
    sizing.generate_dataset(50000, cfg, "cuda")
    import time
    before = runtime.cpu_throttle_counters()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        sizing.generate_dataset(50000, cfg, "cuda")
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    after = runtime.cpu_throttle_counters()
    med = sorted(ts)[len(ts) // 2]
    assert sorted(ts)[-2] < 2.0 * med and max(ts) < 4.0 * med, ts
    if before and after:
        assert after["nr_throttled"] == before["nr_throttled"], (before, after, ts)
