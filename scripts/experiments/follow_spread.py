"""Do the hand-written training paths draw from the SAME stochastic process as the framework path -- or is there a bias?

K seeds x {every path of a surrogate} on one box, same dataset, same initial weights and batch order per seed; prints every curve, then
mean +- sigma of the final training / validation loss per path and the difference of the means in units of its standard error.
Two configurations per surrogate: the reference's (dropout + noise on: the paths use different random streams, so only distributions
can agree) and a deterministic one (dropout 0, noise 0, fixed batch order: the curves themselves must agree to bf16 rounding).

    python scripts/follow_spread.py [pinn|tfd|both] [K] [epochs]
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from openpystruct_amd import dataprep, pinn_fused, sizing, surrogates as S, tfd_fused, train

which = sys.argv[1] if len(sys.argv) > 1 else "both"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
EPOCHS = int(sys.argv[3]) if len(sys.argv) > 3 else 8

rec = sizing.generate_dataset(6000, sizing.SizingConfig(max_e=60), "cuda")


def set_mode(kind, mode):
    if kind == "pinn":
        pinn_fused.ENABLED = mode == "blocks"
        S._FUSED_TAILS = mode != "framework"
        os.environ["OPS_AMD_FUSED_PREP"] = "0" if mode == "framework" else "1"
    else:
        tfd_fused.ENABLED = mode == "fast"
        os.environ["OPS_AMD_FUSED_PREP"] = "1"


def no_diffusion_noise(model):
    model.diffusion._acp.fill_(1.0)          # sqrt(1 - alpha_cumprod) = 0: x_noisy = x whatever the draws


def run(kind, modes, deterministic):
    d = dataprep.prepare(rec, kind=kind, device="cuda")
    n_tr = int(d.X_train.shape[0])
    cfg = (train.PinnConfig if kind == "pinn" else train.TfdConfig)()
    kw = {}
    if deterministic:
        cfg.dropout_rate, cfg.sigma_0 = 0.0, 0.0
        if kind == "tfd":
            kw["init_fn"] = no_diffusion_noise
    res = {m: {"train": [], "val": []} for m in modes}
    for seed in range(1, K + 1):
        order = {}

        def batch_order(epoch, seed=seed):
            if epoch not in order:
                order[epoch] = torch.randperm(n_tr, generator=torch.Generator().manual_seed(1000 * seed + epoch))
            return order[epoch]

        for m in modes:
            set_mode(kind, m)
            out = train.train_surrogate(kind, d, cfg, device="cuda", max_epochs=EPOCHS, seed=seed, batch_order=batch_order, **kw)
            h = out["history"]
            res[m]["train"].append(h["train"]); res[m]["val"].append(h["val"])
            print(f"{kind} {'det' if deterministic else 'sto'} seed {seed:2d} {m:9s} train", " ".join("%.4f" % v for v in h["train"]),
                  "| val", " ".join("%.4f" % v for v in h["val"]), flush=True)
    tag = f"{kind} {'deterministic' if deterministic else 'stochastic'}"
    ref = modes[-1]
    for key in ("train", "val"):
        a_ref = np.array(res[ref][key])
        for m in modes:
            a = np.array(res[m][key])
            fin = a[:, -1]
            line = f"{tag} {key:5s} {m:9s} final mean {fin.mean():.5f} sigma {fin.std(ddof=1):.5f} ({100 * fin.std(ddof=1) / fin.mean():.2f} %)"
            if m != ref:
                diff = fin.mean() - a_ref[:, -1].mean()
                se = np.sqrt(fin.var(ddof=1) / len(fin) + a_ref[:, -1].var(ddof=1) / len(fin))
                paired = a[:, -1] / a_ref[:, -1] - 1.0
                line += (f" | vs {ref}: {100 * diff / a_ref[:, -1].mean():+.2f} % = {diff / max(se, 1e-30):+.2f} standard errors; per-seed ratio - 1: "
                         f"mean {100 * paired.mean():+.2f} % max |.| {100 * np.abs(paired).max():.2f} %; worst epoch-wise |ratio - 1| "
                         f"{100 * np.abs(a / a_ref - 1.0).max():.2f} %")
            print(line, flush=True)


if which in ("pinn", "both"):
    for det in (False, True):
        run("pinn", ["blocks", "tails", "framework"], det)
if which in ("tfd", "both"):
    for det in (False, True):
        run("tfd", ["fast", "framework"], det)
