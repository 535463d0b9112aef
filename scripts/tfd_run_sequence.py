import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from openpystruct_amd import dataprep, sizing, tfd_fused, train
rec = sizing.generate_dataset(6000, sizing.SizingConfig(max_e=60), "cuda")
d = dataprep.prepare(rec, kind="tfd", device="cuda")
seq = sys.argv[1].split(",")
for item in seq:
    if item == "empty":
        torch.cuda.empty_cache(); continue
    tfd_fused.ENABLED = item.startswith("fast")
    tfd_fused.LAYER_FWD = "nolayer" not in item
    tfd_fused.DRAW = "nodraw" not in item
    out = train.train_surrogate("tfd", d, device="cuda", max_epochs=6, seed=1)
    h = out["history"]["train"]
    print(item, "OK " if np.all(np.isfinite(h)) else "NAN", ["%.4f" % v for v in h], flush=True)
