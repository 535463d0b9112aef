"""Worker of tests/test_gpu_dp_rccl.py: started by `python -m torch.distributed.run --nproc-per-node 1` (the launcher touches no GPU),
it creates a ONE-rank `nccl` (= RCCL) process group and runs the surrogates' training loop through the real data-parallel branch
(OPS_AMD_FORCE_DP): flat gradient all-reduce between graph A and graph B, its async form, the one-graph capture of the collective with
its fallback, the dp_segments event record -- next to the plain single-process run of the same seed.  Prints one JSON line."""
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")       # an entry point: before the first HIP call (runtime.configure)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    kinds = sys.argv[1:] or ["pinn", "tfd"]
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    from openpystruct_amd import dataprep, runtime, sizing, train
    rt = runtime.configure()
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "runtime": rt, "runs": {}}
    t = torch.arange(8, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    dist.barrier(device_ids=[local])
    out["allreduce_identity"] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)))
    rec = sizing.generate_dataset(1800, sizing.SizingConfig(max_e=30), dev, seed=11)      # 300 groups -> 240 training groups
    logs = []
    default_one_graph = bool(train._DP_ONE_GRAPH)          # the module's default (r06: the collective captured, one graph per step)
    out["default_one_graph"] = default_one_graph
    for kind in kinds:
        d = dataprep.prepare(rec, kind=kind, seed=0, device=dev, distributed=True)       # scaler moments through the (one-rank) all-reduce
        cfg = {"pinn": train.PinnConfig, "tfd": train.TfdConfig}[kind](batch_size=64)   # 240 rows: 3 full batches + a tail of 48
        runs = {}
        variants = [("plain", dict(force=False)), ("plain_again", dict(force=False)), ("dp_async", dict(force=True, async_=True)),
                    ("dp_blocking", dict(force=True, async_=False)), ("dp_one_graph", dict(force=True, async_=True, one_graph=True)),
                    ("dp_profile", dict(force=True, async_=True, profile=True)), ("dp_default", dict(force=True, async_=True, one_graph=default_one_graph))]
        for name, v in variants:
            train._FORCE_DP = v["force"]
            train._DP_ASYNC = v.get("async_", True)
            train._DP_ONE_GRAPH = v.get("one_graph", False)
            train._DP_PROFILE = v.get("profile", False)
            r = train.train_surrogate(kind, d, cfg, device=dev, max_epochs=3, seed=5, log=logs.append)
            runs[name] = {"train": [float(x) for x in r["history"]["train"]], "val": [float(x) for x in r["history"]["val"]],
                          "r2_val_I": float(r["r2_val_I"]), "dp_segments": r.get("dp_segments"), "dp_mode": r.get("dp_mode")}
        if kind == "tfd":       # r06: library option "deterministic" -- fixed-order reductions, so the one-rank identity is bitwise for this model too
            from openpystruct_amd import _cabi
            _cabi.set_option("deterministic", 1)
            try:
                for name, v in (("plain_det", dict(force=False)), ("dp_det", dict(force=True, one_graph=default_one_graph))):
                    train._FORCE_DP, train._DP_ASYNC, train._DP_ONE_GRAPH, train._DP_PROFILE = v["force"], True, v.get("one_graph", False), False
                    r = train.train_surrogate(kind, d, cfg, device=dev, max_epochs=3, seed=5, log=logs.append)
                    runs[name] = {"train": [float(x) for x in r["history"]["train"]], "val": [float(x) for x in r["history"]["val"]],
                                  "r2_val_I": float(r["r2_val_I"]), "dp_segments": r.get("dp_segments"), "dp_mode": r.get("dp_mode")}
            finally:
                _cabi.set_option("deterministic", 0)
        out["runs"][kind] = runs
    out["log"] = logs
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()
    print("DP_ONE_RANK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
