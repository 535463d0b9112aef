#!/usr/bin/env python3
"""Epoch time of the PINN / Transformer-Diffusion training loops (second half of BASELINE.json's metric).

    python scripts/train_epoch_bench.py --kind pinn                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           scripts/train_epoch_bench.py --kind tfd

Dataset: `--cases` synthetic cases (BASELINE config 3: 50 000) generated ON the GPUs by the HIP sizing
path, each rank its shard (weak scaling of the generator; the training set is the rank's shard, so the
global batch is world x per-GPU batch as in BASELINE config 3).  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import dataprep, sizing, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="pinn", choices=["pinn", "tfd"])
    ap.add_argument("--cases", type=int, default=50000)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--gen-epochs", type=int, default=600, help="max_e of the sizing loop used to generate the data")
    ap.add_argument("--deterministic", action="store_true", help="library option deterministic = 1 (fixed-order reductions in the TFD gradient launches)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lr = int(os.environ.get("LOCAL_RANK", "0"))
    lr %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(lr)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("OPS_AMD_BENCH_BACKEND", "nccl")   # gloo: dry-run of the N > 1 path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", lr)
    if a.deterministic:
        from openpystruct_amd import _cabi
        _cabi.set_option("deterministic", 1)
    t0 = time.perf_counter()
    rec = sizing.generate_dataset(a.cases * world, sizing.SizingConfig(max_e=a.gen_epochs), dev, rank=rank, world=world)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    solves = int(rec["epochs_run"].sum())
    d = dataprep.prepare(rec, kind=a.kind, device=dev, distributed=world > 1)
    out = train.train_surrogate(a.kind, d, device=dev, max_epochs=a.epochs, autocast_dtype=(None if os.environ.get("AC", "1") == "0" else torch.bfloat16), log=(lambda m: print(m, file=sys.stderr)) if rank == 0 else None)
    ep = out["history"]["epoch_s"][1:] or out["history"]["epoch_s"]
    if rank == 0:
        print(json.dumps({"metric": f"{a.kind} epoch time", "value": sum(ep) / len(ep), "unit": "s", "n_gpus": world,
                          "higher_is_better": False, "scaling": "weak", "dtype": "bf16",
                          "config": {"workload": f"{a.cases} generated cases per GPU -> {d.X_train.shape[0]} train groups per GPU, "
                                                 f"batch {train.PinnConfig().batch_size if a.kind == 'pinn' else train.TfdConfig().batch_size} per GPU",
                                     "steps_per_epoch": out["steps_per_epoch"]},
                          "generation": {"seconds": t_gen, "cases_per_gpu": a.cases, "fe_solves_per_gpu": solves,
                                         "mean_epochs_per_case": solves / max(1, a.cases)}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
