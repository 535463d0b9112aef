import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_gpu_pinn_fused as T
for B, p in ((16, 0.2), (33, 0.3), (128, 0.5)):
    for seed in range(40, 52):
        before, crit, x, y, masks, loss, got = T._run_engine(B, p, seed, slope=1.0)
        ref, _, loss_ref = T._reference(before, crit, x, y, masks, p)
        gref = {n: q.grad for n, q in ref.named_parameters()}
        worst = {n: float((got[n] - gref[n]).norm()) / (max(float(gref[n].norm()), T._floor2(n, gref)) + 1e-30) for n in got}
        k = max(worst, key=worst.get)
        print(B, p, seed, "worst %.4f %s" % (worst[k], k), "second %.4f" % sorted(worst.values())[-2], flush=True)
