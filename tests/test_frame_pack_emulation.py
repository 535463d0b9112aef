"""The packed frame kernel's lane protocol (openpystruct_amd/csrc/frame_pack.hpp) restated lane by lane in numpy and run on the CPU: row R in lane
R mod P with entry A[R][C] in register C mod W, ONE line entry per lane and step (index 0: the finished row's right-hand side, 1 + rel: the column
entry, unmasked), pivot / multipliers / right-hand side read back from the line one step ahead, the forward substitution one column behind the
factorisation, the column of L stored unconditionally (slot W - 1 for lanes outside the window), w flushed at row-group boundaries, and the
backward sweep in passes of P / 8 columns with eight lanes per column and the pass's triangle solved redundantly.

What this covers without a GPU: the index logic -- window masks at the end of the matrix, rows entering in groups of G while KG + G rows are in
flight, stale register contents of finished rows and of rows ahead of the window meeting only zero multipliers or dead slots, equation counts
that are no multiple of anything -- for every compiled (W, P, G).  The HIP kernel itself is held to the oracle in tests/test_gpu_frames.py."""
import numpy as np
import pytest

CONFIGS = [(6, 16, 4), (10, 16, 4), (12, 16, 4), (16, 32, 8), (18, 32, 8), (22, 32, 8), (24, 32, 8), (28, 32, 4), (30, 32, 2)]      # fp_config of frame_pack.hpp
MAX_KD = {6: 5, 10: 9, 12: 11, 16: 15, 18: 17, 22: 21, 24: 23, 28: 27, 30: 29}


def spd_band(rng, n, kd):
    """A = L D L^T with a random unit lower band factor: symmetric positive definite, half bandwidth kd."""
    L = np.eye(n)
    for r in range(n):
        for c in range(max(0, r - kd), r):
            L[r, c] = rng.uniform(-0.4, 0.4)
    d = rng.uniform(0.5, 2.0, size=n)
    return (L * d) @ L.T


def emulate(A, b, kd, W, P, G, garbage=None):
    """One frame through the packed kernel's steps; returns (x, bad)."""
    n = A.shape[0]
    KG = (kd // G + 1) * G
    assert KG + G <= P and W > kd and W % 2 == 0
    lane = np.arange(P)
    reg = np.zeros((P, W)) if garbage is None else garbage.uniform(-3, 3, size=(P, W))      # (a wave's registers are never clean)
    y = np.zeros(P); lp = np.zeros(P); w = np.zeros(P)
    line = np.zeros((2, P))
    xs = np.zeros(n + P)
    xs[:n] = b                                                # the right-hand side is staged in xs until a row's group is built
    Lc = np.full((n + 3) * W, np.nan)                         # what is never stored must never be read unmasked
    bad = False

    def take_group(g0):                                       # rows g0 .. g0 + G - 1 (rows past n: zero rows) into their lanes
        for s in range(G):
            R = g0 + s
            r = R % P
            reg[r, :] = 0.0
            if R < n:
                for C in range(max(0, R - kd), R + 1):
                    reg[r, C % W] = A[R, C]
                y[r] = xs[R]
            else:
                reg[r, R % W] = 1.0                           # (the plan's unit-diagonal rows between n and the end of its group; beyond: zeros)
                y[r] = 0.0

    for g0 in range(0, KG + G, G):
        take_group(g0)
    # line of column 0
    idx = (lane + 1) % P
    lim0 = min(kd + 1, n)
    line[0, idx] = np.where((idx >= 1) & (idx - 1 < lim0), reg[:, 0], 0.0)
    zp, d = 0.0, line[0, 1]
    rd = 1.0 / d
    bad |= not (d > 0)
    a1, a2 = line[0, 2], line[0, 3]
    steps = (n + 1) // 2 * 2
    for j in range(steps):
        S = j % W
        if j > 0 and j % G == 0 and j < n:                    # boundary: rows [j - G, j) are finished, rows [j + KG, j + KG + G) enter
            slot = (lane - (j - G)) % P
            for r in lane[slot < G]:
                xs[j - G + slot[r]] = w[r]
            take_group(j + KG)
        rel = (lane - j) % P
        below = max(n - 1 - j, 0)
        lim = min(kd, below)
        inwin = (rel >= 1) & (rel <= lim)
        y[:] = y - lp * zp
        a = reg[:, S].copy()
        l = np.where(inwin, a * rd, 0.0)
        lp[:] = l
        reg[:, (S + 1) % W] -= l * a1
        line[(j + 1) & 1, rel] = np.where(rel == 0, y, reg[:, (S + 1) % W])          # unmasked
        col = np.where(inwin, rel - 1, W - 1)
        Lc[j * W + col] = l
        if j < n:
            w[rel == 0] = y[rel == 0] * rd
        cb = line[j & 1]
        if W > 2:
            reg[:, (S + 2) % W] -= l * a2
        for t in range(3, W):
            reg[:, (S + t) % W] -= l * cb[t + 1]
        nb = line[(j + 1) & 1]
        zp, d = nb[0], nb[1]
        with np.errstate(divide="ignore", invalid="ignore"):
            rd = 1.0 / d
        if j + 1 < n:
            bad |= not (d > 0)
        a1, a2 = nb[2], nb[3]
        assert np.isfinite(reg).all() and np.isfinite(y).all(), j          # nothing outside the window may turn a live slot non-finite
    Rl = n - 1 - ((n - 1 - lane) % P)
    for r in lane[Rl >= 0]:
        xs[Rl[r]] = w[r]
    # ---- backward sweep ----
    U, MF = P // 8, (W + 7) // 8
    xs[n:n + P] = 0.0
    u, k = lane >> 3, lane & 7
    jb = n - 1
    while jb >= 0:
        ju = jb - u
        jc = np.maximum(ju, 0)
        kdj = np.where(ju >= 0, np.minimum(kd, n - 1 - ju), 0)
        acc = np.zeros(P)
        for m in range(MF):
            relm = k + 1 + 8 * m
            far = (relm <= kdj) & (relm > u)
            e = np.minimum(k + 8 * m, W - 1)
            lf = Lc[jc * W + e]
            acc += np.where(far, lf, 0.0) * xs[jc + relm]
        acc = np.where(np.isnan(acc), np.inf, acc)             # (a NaN here means an unstored slot was used unmasked)
        s8 = acc.reshape(U, 8).sum(axis=1)                    # the eight lanes of a column
        t = xs[np.maximum(jb - np.arange(U), 0)] - s8
        tri = {}
        for wv in range(1, U):                                # lanes (u = wv, k < wv) publish L[jb - v][jb - wv], v = wv - 1 - k
            jw = jb - wv
            kdw = min(kd, n - 1 - jw) if jw >= 0 else 0
            for kk in range(wv):
                v = wv - 1 - kk
                tri[(wv, v)] = Lc[max(jw, 0) * W + kk] if kk + 1 <= kdw else 0.0
        x = t.copy()
        for wv in range(1, U):
            for v in range(wv):
                x[wv] -= tri[(wv, v)] * x[v]
        for rr in range(U):
            if 0 <= jb - rr < n:
                xs[jb - rr] = x[rr]
        jb -= U
    return xs[:n].copy(), bad


@pytest.mark.parametrize("W,P,G", CONFIGS)
def test_packed_lane_protocol_solves_band_systems(W, P, G):
    rng = np.random.default_rng(W * 100 + P)
    kd_max = MAX_KD[W]
    for n, kd in [(kd_max + 1, kd_max), (P - 3, kd_max), (P + 1, kd_max), (3 * P + 5, kd_max), (2 * P + G, max(1, kd_max - 2)), (7, min(kd_max, 3)),
                  (90, kd_max), (37, max(2, kd_max // 2))]:
        kd = min(kd, n - 1)
        A = spd_band(rng, n, kd)
        b = rng.uniform(-1, 1, size=n)
        x, bad = emulate(A, b, kd, W, P, G, garbage=rng)
        ref = np.linalg.solve(A, b)
        assert not bad, (n, kd)
        assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max(), (W, P, G, n, kd, np.abs(x - ref).max())


def test_packed_lane_protocol_reports_a_non_positive_pivot():
    rng = np.random.default_rng(5)
    A = spd_band(rng, 40, 11)
    A[17, 17] -= 50.0                                         # not positive definite from equation 17 on
    _, bad = emulate(A, rng.uniform(-1, 1, size=40), 11, 12, 16, 4)
    assert bad
