"""Epoch time of the data-parallel training step on ONE rank of RCCL: two graphs around an eager all-reduce against the one-graph capture
of the collective, next to the plain single-process step.  Start with the launcher:
    OPS_AMD_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 scripts/dp_one_rank_timing.py"""
import json, os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

for k_, v_ in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29711"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k_, v_)          # (without the launcher: the one-rank environment env:// needs)
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
from openpystruct_amd import dataprep, runtime, sizing, train
runtime.configure()
rec = sizing.generate_dataset(50000, sizing.SizingConfig(), dev)
out = {}
for kind in ("pinn", "tfd"):
    d = dataprep.prepare(rec, kind=kind, device=dev, distributed=True)
    for name, force, one in (("plain", False, False), ("dp_two_graphs", True, False), ("dp_one_graph", True, True)):
        train._FORCE_DP, train._DP_ONE_GRAPH = force, one
        r = train.train_surrogate(kind, d, device=dev, max_epochs=8, seed=3)
        ep = r["history"]["epoch_s"][2:]
        out[f"{kind}/{name}"] = {"epoch_ms": round(1e3 * sum(ep) / len(ep), 3), "step_us": round(1e6 * sum(ep) / len(ep) / r["steps_per_epoch"], 1),
                                 "mode": (r.get("dp_mode") or {}).get("step")}
        print(kind, name, out[f"{kind}/{name}"], flush=True)
dist.destroy_process_group()
print("DP_TIMING " + json.dumps(out))
summ = {k: {f"{n}_step_us": out[f"{k}/{n}"]["step_us"] for n in ("plain", "dp_two_graphs", "dp_one_graph")} for k in ("pinn", "tfd")}
summ["what"] = ("epoch time / steps (validation pass included) of the surrogates' training loop on 50 000 generated cases, epochs 3-8, one rank of RCCL "
                "(scripts/dp_one_rank_timing.py): plain single-process step, two graphs around an eager all-reduce, ONE graph incl. the all-reduce (r06 default)")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(summ, open("gpurun_out/r06_dp_one_rank_timing.json", "w"), indent=1)
