#!/usr/bin/env python3
"""Wall time of make_cases / optimize_cases / generate_dataset at 50 000 and 200 000 cases (warm and cold)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import sizing
cfg = sizing.SizingConfig()
def T(f, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); return r, time.perf_counter() - t
for n in (50000, 200000):
    for rep in range(2):
        cases, t1 = T(sizing.make_cases, n, cfg, 20250307, device="cuda")
        st, t2 = T(sizing.optimize_cases, cases, cfg, "cuda")
        rec, t3 = T(sizing.generate_dataset, n, cfg, "cuda")
        print(n, "make_cases %.3f optimize %.3f generate_dataset(total) %.3f  epochs max %d mean %.1f" % (t1, t2, t3, int(st.epochs_run.max()), float(st.epochs_run.float().mean())))
