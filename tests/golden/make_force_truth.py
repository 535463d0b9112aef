"""Generates tests/golden/force_truth.npz: beams with a 50-DIGIT solution (mpmath banded elimination + the end-force
recovery of SURVEY Appendix A.4 evaluated in 50 digits), so that the error of the HIP kernel and the error of the
double-precision band solver the reference uses (`system('BandSPD')` = LAPACK dpbsv, SingleCore.py:120; restated in
oracle/beam_oracle.py) can be measured SEPARATELY.  The parity tests then require the kernel's error against the truth
to stay within 10x of the band solver's own error against the truth, for displacements AND element end forces, on every
tiling (tests/test_force_truth.py) -- the bar VERDICT r01 set for the end-force parity hole.

Inputs: the reference bridge (SingleCore.py:58-62) with "trajectory" and adversarial inertias (first beams of the committed
bridge_*.npz fixtures), and long beams Ne in {127, 255, 511, 1023} (the P = 32 / 64 tilings) with six supports.
Run:  python tests/golden/make_force_truth.py      (about a minute; needs mpmath, no reference access)
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import beam_oracle as bo  # noqa: E402

mp.mp.dps = 50


def mp_truth(x, E, I, fix, Fy, wy):
    """Dense-band Gaussian elimination (SPD, no pivoting, half bandwidth 3) and force recovery in 50 digits."""
    N = len(x); Ne = N - 1; n = 2 * N
    f = lambda a: mp.mpf(float(a))  # noqa: E731
    band = [dict() for _ in range(n)]          # row -> {col: value}, |row - col| <= 3
    rhs = [mp.mpf(0)] * n
    for i in range(N):
        rhs[2 * i] += f(Fy[i])
    for e in range(Ne):
        L = f(x[e + 1]) - f(x[e]); EI = f(E) * f(I[e]); w = f(wy)
        k = [[12 * EI / L**3, 6 * EI / L**2, -12 * EI / L**3, 6 * EI / L**2],
             [6 * EI / L**2, 4 * EI / L, -6 * EI / L**2, 2 * EI / L],
             [-12 * EI / L**3, -6 * EI / L**2, 12 * EI / L**3, -6 * EI / L**2],
             [6 * EI / L**2, 2 * EI / L, -6 * EI / L**2, 4 * EI / L]]
        fe = [w * L / 2, w * L * L / 12, w * L / 2, -w * L * L / 12]
        for a in range(4):
            rhs[2 * e + a] += fe[a]
            for b in range(4):
                band[2 * e + a][2 * e + b] = band[2 * e + a].get(2 * e + b, mp.mpf(0)) + k[a][b]
    for i in range(N):
        for bit, d in ((1, 2 * i), (2, 2 * i + 1)):
            if int(fix[i]) & bit:              # constraints('Plain'): identity row / column, zero right-hand side
                for j in range(max(0, d - 3), min(n, d + 4)):
                    band[d][j] = mp.mpf(0); band[j][d] = mp.mpf(0)
                band[d][d] = mp.mpf(1); rhs[d] = mp.mpf(0)
    for c in range(n):
        piv = band[c][c]
        for i in range(c + 1, min(n, c + 4)):
            m = band[i].get(c, mp.mpf(0)) / piv
            if m == 0:
                continue
            for j in range(c, min(n, c + 4)):
                band[i][j] = band[i].get(j, mp.mpf(0)) - m * band[c].get(j, mp.mpf(0))
            rhs[i] -= m * rhs[c]
    u = [mp.mpf(0)] * n
    for i in range(n - 1, -1, -1):
        s = rhs[i]
        for j in range(i + 1, min(n, i + 4)):
            s -= band[i].get(j, mp.mpf(0)) * u[j]
        u[i] = s / band[i][i]
    V, M = [], []
    for e in range(Ne):
        L = f(x[e + 1]) - f(x[e]); EI = f(E) * f(I[e]); w = f(wy)
        chord = (u[2 * e + 2] - u[2 * e]) / L
        p1, p2 = u[2 * e + 1] - chord, u[2 * e + 3] - chord
        q1 = 4 * EI / L * p1 + 2 * EI / L * p2 - w * L * L / 12
        q2 = 2 * EI / L * p1 + 4 * EI / L * p2 + w * L * L / 12
        V.append((q1 + q2) / L - w * L / 2); M.append(q1)
    tof = lambda a: np.array([float(t) for t in a])  # noqa: E731
    return tof(u[0::2]), tof(u[1::2]), tof(V), tof(M)


def long_beam(rng, Ne, inertia):
    N = Ne + 1
    x = np.concatenate([[0.0], np.cumsum(rng.uniform(1.5, 2.5, Ne))])       # non-uniform mesh
    fix = np.zeros(N, np.uint8); fix[0] = 1
    for r in (0.1, 0.3, 0.7, 0.85, 0.99):
        fix[int(r * Ne)] = 1
    lo = 3e-3 if inertia == "trajectory" else 1e-8
    I = np.exp(rng.uniform(np.log(lo), np.log(0.75 if inertia == "trajectory" else 0.5), Ne))
    Fy = np.zeros(N)
    idx = rng.choice(np.nonzero(fix == 0)[0][:-1], 4, replace=False)
    Fy[idx] = rng.uniform(bo.MAX_FORCE, bo.MIN_FORCE, 4)
    return x, I, fix, Fy


def main():
    out, names = {}, []
    rng = np.random.default_rng(20250311)

    def add(name, x, I, fix, Fy):
        v, th, V, M = mp_truth(x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
        K, _ = bo.assemble_beam(x, bo.E_REF, I, Fy, bo.UDL_REF)
        free = np.ones(K.shape[0], bool); free[0::2] = (fix & 1) == 0; free[1::2] = (fix & 2) == 0
        Kf = K[np.ix_(free, free)]
        d = 1.0 / np.sqrt(np.diag(Kf))
        ks = np.array(np.linalg.cond(Kf * d[:, None] * d[None, :]))    # Jacobi-scaled condition number (van der Sluis)
        for k, a in (("x", x), ("I", I), ("fix", fix), ("Fy", Fy), ("v", v), ("theta", th), ("V", V), ("M", M), ("kscaled", ks)):
            out[f"{name}/{k}"] = a
        names.append(name)
        print(name, flush=True)

    for fx, tag, nb in (("bridge_trajectory.npz", "bridge_traj", 3), ("bridge_adversarial.npz", "bridge_adv", 5)):
        z = np.load(os.path.join(HERE, fx))
        for b in range(nb):
            add(f"{tag}{b}", z["x"], z["I"][b], z["fix"], z["Fy"][b])
    for Ne in (127, 255, 511, 1023):
        for inertia in ("trajectory", "adversarial"):
            for b in range(2):
                add(f"long{Ne}_{inertia[:4]}{b}", *long_beam(rng, Ne, inertia))
    out["names"] = np.array(names)
    out["E"] = np.array(bo.E_REF); out["wy"] = np.array(bo.UDL_REF)
    np.savez_compressed(os.path.join(HERE, "force_truth.npz"), **out)


if __name__ == "__main__":
    main()
