import os, sys
os.environ["OPS_AMD_TFD_TRACE_BWD"] = "1"
os.environ["OPS_AMD_GRAPH"] = "0"          # eager steps: every launch gets its own trace buffer
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from openpystruct_amd import dataprep, sizing, tfd_fused as TF, train
rec = sizing.generate_dataset(12000, sizing.SizingConfig(max_e=60), "cuda")
d = dataprep.prepare(rec, kind="tfd", device="cuda")
out = train.train_surrogate("tfd", d, device="cuda", max_epochs=3, seed=1)
torch.cuda.synchronize()
names = ["entry", "loads issued", "rows staged", "LN2 bwd + d_f", "d_h + act bwd", "d_y1 + LN1 bwd", "d_ctx", "attention bwd", "end"]
trs = [t.cpu().numpy().reshape(-1, 16) for t in TF._TRACE_BWD if t.numel() == 16 * 256]
print("launches traced", len(trs))
for which, sel in (("last layer (g16 only)", trs[-8::2]), ("first layer (g32 only)", trs[-7::2])):
    med = np.median(np.stack([np.median((t[:, :9] - t[:, 0].min()) / 100.0, axis=0) for t in sel]), axis=0)
    print(which, " ".join("%s %.1f" % (n, v) for n, v in zip(names, med)))
