"""Model quality at BASELINE config 3 / 4 size: PINN and Transformer-Diffusion trained on 50 000 generated cases
to the reference's early stop (patience 10, PINN:797-806 / TFD:780-789), three seeds each, on the hand-written fast path and on the
framework path (nn.Module forward, autograd, framework losses: every A/B switch off), same data, same seeds.

    python scripts/quality_run.py fast|framework [out.json]

Reports per run: validation R^2 on un-standardised inertias of the reloaded best checkpoint (PINN:815-852 / TFD:800-829), best
validation loss, epochs run, wall time.  One process per path: the switches are read at import."""
import json
import os
import sys
import time

path = sys.argv[1] if len(sys.argv) > 1 else "fast"
if path == "framework":
    # (openpystruct_amd/switches.py: the one variable, parsed at import)
    os.environ["OPS_AMD_SWITCHES"] = "pinn_layer_blocks=0,tfd_fast_encoder=0,shadow_linear=0,fused_loss=0,fused_prep=0,pinn_fused_tails=0,pinn_fused_stencil=0"
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from openpystruct_amd import dataprep, runtime, sizing, train  # noqa: E402

runtime.configure()
t0 = time.perf_counter()
rec = sizing.generate_dataset(50000, sizing.SizingConfig(), "cuda")
torch.cuda.synchronize()
out = {"path": path, "cases": 50000, "generate_s": time.perf_counter() - t0, "runs": []}
for kind in ("pinn", "tfd"):
    d = dataprep.prepare(rec, kind=kind, device="cuda")
    for seed in (1, 2, 3):
        t0 = time.perf_counter()
        r = train.train_surrogate(kind, d, device="cuda", seed=seed)
        torch.cuda.synchronize()
        run = {"kind": kind, "seed": seed, "r2_val_I": float(r["r2_val_I"]), "best_val": float(r["best_val"]), "epochs": int(r["epochs"]),
               "wall_s": time.perf_counter() - t0, "train_groups": int(d.X_train.shape[0]), "val_groups": int(d.X_val.shape[0]),
               "final_train": float(r["history"]["train"][-1]), "mean_epoch_s": float(sum(r["history"]["epoch_s"][1:]) / max(1, len(r["history"]["epoch_s"]) - 1))}
        out["runs"].append(run)
        print(json.dumps(run), flush=True)
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", f"r05_quality_{path}.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
