"""The kernel's per-lane arithmetic (openpystruct_amd/csrc/beam_math.hpp) run lane by lane on
the CPU (tests/csrc/emul_beam.cpp) against the oracle and the golden fixtures, for every
compiled tiling.  Covers the algorithm without a GPU; the GPU parity tests are in test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle import beam_oracle as bo
from tests.helpers import TILINGS, emul_solve, load_golden, relerr

FIT100 = [t for t in TILINGS if t[0] * t[1] >= 101]


@pytest.mark.parametrize("P,M", FIT100)
@pytest.mark.parametrize("name,tol_u,tol_f", [("bridge_uniform", 5e-11, 5e-10), ("bridge_trajectory", 5e-9, 1e-6)])
def test_emulated_lanes_vs_golden(golden_dir, P, M, name, tol_u, tol_f):
    g = load_golden(os.path.join(golden_dir, name + ".npz"))
    v, th, V, Mz, st = emul_solve(P, M, g["x"], g["E"], g["I"], g["fix"], g["Fy"], g["wy"])
    assert (st == 0).all()
    assert relerr(v, g["v"]) < tol_u and relerr(th, g["theta"]) < tol_u
    assert relerr(V, g["V"]) < tol_f and relerr(Mz, g["M"]) < tol_f


@pytest.mark.parametrize("P,M", [(16, 7), (8, 13)])
def test_emulated_random_bridge(golden_dir, P, M):
    g = load_golden(os.path.join(golden_dir, "random_bridge.npz"))
    v, th, V, Mz, st = emul_solve(P, M, g["x"], g["E"], g["I"], g["fix"], g["Fy"], g["wy"])
    assert (st == 0).all()
    assert relerr(v, g["v"]) < 1e-7 and relerr(th, g["theta"]) < 1e-7


@pytest.mark.parametrize("P,M", TILINGS)
@pytest.mark.parametrize("Ne", [1, 2, 3, 7, 13, 14, 50, 100, 103, 127, 255, 511, 1023])
def test_emulated_ragged_sizes(P, M, Ne):
    if P * M < Ne + 1:
        pytest.skip("tiling too small for this Ne")
    if Ne > 127 and P * M > 4 * (Ne + 1):
        pytest.skip("covered by a tighter tiling")
    rng = np.random.default_rng(Ne)
    N = Ne + 1
    x = np.sort(rng.uniform(0, 3.0 * Ne, size=N)) + np.arange(N) * 0.5   # non-uniform spacing
    fix = np.zeros(N, dtype=np.uint8); fix[0] = 1; fix[-1] = 1
    if N > 4:
        fix[N // 3] = 1
    if Ne == 1:
        fix[0] = 3   # one element: clamp it, else it is a mechanism
    I = np.exp(rng.uniform(np.log(1e-2), np.log(0.5), size=(3, Ne)))
    Fy = rng.uniform(-1e5, 0, size=(3, N))
    ref = bo.solve_beam_batched(x, 2.0e11, I, fix, Fy, -750.0)
    out = emul_solve(P, M, x, 2.0e11, I, fix, Fy, -750.0)
    assert (out[4] == 0).all()
    # cond(K_ff) grows like Ne^4 on these random non-uniform meshes (5e8 at Ne=50 ... 1e14 at Ne=1023);
    # both the oracle's band Cholesky and the lane algorithm sit at ~1e-17*cond: cond-aware bound
    K, _ = bo.assemble_beam(x, 2.0e11, I[0], Fy[0], -750.0)
    free = np.ones(2 * N, dtype=bool); free[0::2] = (fix & 1) == 0; free[1::2] = (fix & 2) == 0
    tol = max(1e-10, 2e-16 * np.linalg.cond(K[np.ix_(free, free)]))
    assert relerr(out[0], ref[0]) < tol and relerr(out[1], ref[1]) < tol


@pytest.mark.parametrize("P,M", [(16, 7), (64, 2)])
def test_emulated_singular_flags_status(P, M):
    # a non-positive pivot is what dpbsv / analyze() can detect reliably (a mechanism's last pivot is
    # only ~0 in floating point, in LAPACK too): zero-stiffness beam and negative-inertia beam
    x = np.linspace(0, 10, 11)
    fix = np.zeros(11, dtype=np.uint8); fix[0] = fix[-1] = 1
    I = np.full((3, 10), 0.1); I[0, :] = 0.0; I[1, 4] = -0.1
    Fy = np.zeros((3, 11)); Fy[:, 5] = -1.0
    out = emul_solve(P, M, x, 2e11, I, fix, Fy, 0.0)
    assert out[4][0] != 0 and out[4][1] != 0 and out[4][2] == 0
    assert np.isnan(out[0][:2]).all() and np.isfinite(out[0][2]).all()
