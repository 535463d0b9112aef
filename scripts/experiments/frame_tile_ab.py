"""A/B of the two wave-per-frame mappings (OPS_AMD_FRAME_TILE = 0 / 1): factor columns in the workspace and results, bit for bit."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import frames  # noqa: E402


def fw_width(kd):
    return 16 if kd < 16 else 24 if kd < 24 else 36 if kd < 36 else 52 if kd < 52 else 56


def run(bays, stories, B=3, verbose=True):
    topo = frames.grid_frame(bays, stories)
    rng = np.random.default_rng(bays * 100 + stories)
    I = torch.as_tensor(np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne))), device="cuda")
    n, kd = topo.n_eq, max(topo.kd, 3)
    W = fw_width(kd)
    rows = n + 64 + 16
    out = {}
    for tile in ("0", "1"):
        os.environ["OPS_AMD_FRAME_TILE"] = tile
        sol = frames.frame_solve(topo, I)
        torch.cuda.synchronize()
        ws = list(topo._ws.values())[0]
        per = rows * (W + 1)
        L = ws[: B * per * 8].view(torch.float64).view(B, per)[:, : rows * W].reshape(B, rows, W)[:, :n].clone().cpu().numpy()
        out[tile] = (sol.disp.clone().cpu().numpy(), L, sol.status.clone().cpu().numpy())
    d0, L0, s0 = out["0"]
    d1, L1, s1 = out["1"]
    # only the band part of each column is meaningful
    ok = True
    for j in range(n):
        kdj = min(kd, n - 1 - j)
        if not np.array_equal(L0[:, j, :kdj], L1[:, j, :kdj]):
            bad = np.argwhere(L0[:, j, :kdj] != L1[:, j, :kdj])[0]
            if verbose:
                print(f"  {bays}x{stories} n={n} kd={kd} W={W}: first differing L column {j}, frame {bad[0]}, rel {bad[1] + 1}: {L0[bad[0], j, bad[1]]!r} vs {L1[bad[0], j, bad[1]]!r}")
                print("   col wave:", L0[bad[0], j, :kdj][:12])
                print("   col tile:", L1[bad[0], j, :kdj][:12])
            ok = False
            break
    same = np.array_equal(d0, d1)
    if not same and verbose and ok:
        # w = L^T x of both runs (x from the displacements through the equation numbers): first forward-substitution entry that differs
        neq = topo.d_node_eq.cpu().numpy().reshape(-1)
        for b in range(B):
            xs = []
            for d in (d0, d1):
                x = np.zeros(n)
                flat = d[b].reshape(-1)
                for i, e in enumerate(neq):
                    if e >= 0:
                        x[e] = flat[i]
                w = x.copy()
                for j in range(n):
                    kdj = min(kd, n - 1 - j)
                    w[j] += (L0[b, j, :kdj] * x[j + 1:j + 1 + kdj]).sum()
                xs.append(w)
            rel = np.abs(xs[0] - xs[1]) / (np.abs(xs[0]) + 1e-300)
            badj = np.argwhere(rel > 1e-9).ravel()
            print(f"   frame {b}: w differs at j = {badj[:20]} ... ({len(badj)} of {n}); w0 {xs[0][badj[:4]]} w1 {xs[1][badj[:4]]}")
            break
    print(f"{bays}x{stories} n={n} kd={kd} W={W}: L equal {ok}, disp equal {same}, max |d1-d0| {np.abs(d1 - d0).max():.3e}, status {s0.sum()} {s1.sum()}")
    return ok and same


if __name__ == "__main__":
    cases = [(1, 1), (2, 3), (4, 2), (7, 5), (10, 10), (15, 16), (3, 3), (5, 5), (12, 4), (16, 3)]
    if len(sys.argv) > 1:
        cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    good = all([run(*c) for c in cases])
    print("ALL EQUAL" if good else "MISMATCH")
    sys.exit(0 if good else 1)
