import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import dataprep, sizing, train
rec = sizing.generate_dataset(20000, sizing.SizingConfig(), "cuda")
d = dataprep.prepare(rec, kind="pinn", device="cuda")
out = train.train_surrogate("pinn", d, device="cuda", max_epochs=25, seed=1)
h = out["history"]
print(os.environ.get("OPS_AMD_PINN_FUSED_TAILS", "1"), os.environ.get("OPS_AMD_FUSED_PREP", "1"), "train", [round(v, 4) for v in h["train"][::6]], "val", [round(v, 4) for v in h["val"][::6]], "r2", round(out["r2_val_I"], 4))
