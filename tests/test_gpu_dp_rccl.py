"""The RCCL path on the ONE GPU a test box has: a one-rank `nccl` process group, created by the real launcher,
drives `train_surrogate` through its data-parallel branch and `bench.py` through its barrier / timing all-reduce.  A one-rank
all-reduce is the identity and 1 / world = 1, so the data-parallel runs must reproduce the plain run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port the kernel hands out as free right now (bind to 0), so that two launches of one test process never share a rendezvous port."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(args, env_extra, timeout=600):
    """`python -m torch.distributed.run --nproc-per-node 1 ...` as a CHILD process (never an exec from this GPU-initialised one)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return p.stdout, p.stderr


def test_training_through_the_data_parallel_branch_on_one_rank_of_rccl():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    # (switch pinn_norm_fold = 0: the plain PINN run takes its clip norm from the norm launch like the data-parallel run does behind the
    #  all-reduce -- not from the weight-gradient launch's partial sums, the same sum in another order -- so that "a one-rank all-reduce
    #  is the identity" can be asserted bit for bit)
    out, err = _launch([os.path.join("tests", "dp_one_rank_worker.py")], {"OPS_AMD_SWITCHES": "pinn_norm_fold=0"})
    line = [l for l in out.splitlines() if l.startswith("DP_ONE_RANK ")][-1]
    r = json.loads(line[len("DP_ONE_RANK "):])
    assert r["backend"] == "nccl" and r["world"] == 1 and r["allreduce_identity"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dp_one_rank_rccl.json"), "w") as f:
        json.dump(r, f, indent=1)
    for kind, runs in r["runs"].items():
        plain = np.array(runs["plain"]["train"] + runs["plain"]["val"])
        again = np.array(runs["plain_again"]["train"] + runs["plain_again"]["val"])
        assert np.isfinite(plain).all()
        spread = float(np.abs(plain - again).max() / np.abs(plain).max())
        for name in ("dp_async", "dp_blocking", "dp_one_graph", "dp_profile", "dp_default"):
            got = np.array(runs[name]["train"] + runs[name]["val"])
            diff = float(np.abs(got - plain).max() / np.abs(plain).max())
            if kind == "pinn":
                # the PINN step's launches are deterministic (no float atomics): a one-rank all-reduce is the identity, bit for bit,
                # in every form of the step
                assert spread == 0.0 and diff == 0.0, (kind, name, diff, spread)
            else:
                # the TFD step adds its split-row weight-gradient tiles with float atomics: two runs of ONE configuration land on one of
                # a few nearby trajectories (ten suite runs: differences of 0 or ~1.3e-4 between any two runs, plain or data parallel), so
                # the statement is a tolerance -- far below what a wrong average / missing all-reduce would show (O(1))
                assert diff <= max(20 * spread, 2e-3), (kind, name, diff, spread)
        if kind == "tfd":       # deterministic mode (library option, r06): the one-rank all-reduce is the identity BIT FOR BIT for this model too
            pd = np.array(runs["plain_det"]["train"] + runs["plain_det"]["val"])
            dd = np.array(runs["dp_det"]["train"] + runs["dp_det"]["val"])
            assert np.isfinite(pd).all() and np.array_equal(pd, dd), float(np.abs(pd - dd).max())
            assert runs["dp_det"]["dp_mode"]["step"] == "one graph incl. the all-reduce"
        assert runs["plain"]["dp_mode"] is None
        # r06: the DEFAULT data-parallel step under nccl is the one-graph form (the collective captured; first replay under the stall timer)
        assert r["default_one_graph"] and runs["dp_default"]["dp_mode"]["step"] == "one graph incl. the all-reduce"
        for name in ("dp_async", "dp_blocking", "dp_one_graph", "dp_profile"):
            m = runs[name]["dp_mode"]
            assert m["backend"] == "nccl" and m["world"] == 1 and m["forced_one_rank"] and m["async"] == (name != "dp_blocking")
            assert m["step"] in ("graph A | all-reduce | graph B", "one graph incl. the all-reduce") and (name == "dp_one_graph" or m["step"][0] == "g")
        seg = runs["dp_profile"]["dp_segments"]
        assert seg is not None and seg["world"] == 1 and seg["graph_a_us"] > 0 and seg["allreduce_us"] >= 0 and seg["graph_b_us"] > 0


def test_bench_line_through_the_rccl_barrier_and_timing_allreduce_on_one_rank():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    out, err = _launch(["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras", "--train-epochs", "2"],
                       {"OPS_AMD_FORCE_DP": "1"})
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 1e8 and rec["roofline"]["frac"] > 0.2
    se = rec["surrogate_epochs"]
    assert "error" not in se, se
    for kind in ("pinn", "tfd"):
        assert se[kind]["epoch_s"] > 0 and se[kind]["dp_segments"]["world"] == 1 and se[kind]["dp_segments"]["one_graph"]      # the data-parallel step ran: ONE graph incl. the all-reduce
    with open(os.path.join(ROOT, "gpurun_out", "bench_one_rank_rccl.json"), "w") as f:
        json.dump(rec, f)
