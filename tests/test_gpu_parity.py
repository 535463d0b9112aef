"""GPU parity tests proper: the HIP kernel, called through the C ABI, against the oracle on the
same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full size
(10 000 beams x 100 elements) -- through size-independent properties of the linear FE problem.

Tolerances (relative to each beam's max |.|): north_star asks 1e-6 on displacements; the tests
hold the kernel to the oracle's own eps*cond level, which is far tighter for the reference's
inertia ranges."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import beam_oracle as bo  # noqa: E402
from oracle import c_oracle as co  # noqa: E402
from tests.helpers import FAT_P, TILINGS, kappa_scaled, load_golden, relerr  # noqa: E402


@pytest.fixture(scope="module")
def oa():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import openpystruct_amd as oa_
    from openpystruct_amd import _cabi

    _cabi.load()   # fail loudly if the HIP extension is missing
    return oa_


def _gpu(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def _solve(oa, x, E, I, fix, Fy, wy, tiling=0):
    out = oa.beam_solve(_gpu(x), _gpu(E), _gpu(I), _gpu(fix, torch.uint8), _gpu(Fy), _gpu(wy), tiling=tiling)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out]


P_OF_100 = sorted({p for p, m in TILINGS if p * m >= 101})


@pytest.mark.parametrize("tiling", [0] + P_OF_100)
@pytest.mark.parametrize("name,tol_u,tol_f", [("bridge_uniform", 1e-10, 1e-9), ("bridge_trajectory", 1e-8, 2e-6)])
def test_golden_fixed_bridge(oa, golden_dir, tiling, name, tol_u, tol_f):
    g = load_golden(os.path.join(golden_dir, name + ".npz"))
    v, th, V, M, st = _solve(oa, g["x"], g["E"], g["I"], g["fix"], g["Fy"], g["wy"], tiling)
    assert (st == 0).all()
    assert relerr(v, g["v"]) < tol_u and relerr(th, g["theta"]) < tol_u
    assert relerr(V, g["V"]) < tol_f and relerr(M, g["M"]) < tol_f


def test_golden_random_bridge_per_beam_geometry(oa, golden_dir):
    g = load_golden(os.path.join(golden_dir, "random_bridge.npz"))
    v, th, V, M, st = _solve(oa, g["x"], g["E"], g["I"], g["fix"], g["Fy"], g["wy"])
    assert (st == 0).all()
    assert relerr(v, g["v"]) < 1e-7 and relerr(th, g["theta"]) < 1e-7
    assert relerr(V, g["V"]) < 1e-5 and relerr(M, g["M"]) < 1e-5


@pytest.mark.parametrize("tiling", [0] + P_OF_100)
def test_golden_adversarial_displacements_and_forces(oa, golden_dir, tiling):
    """I in [1e-8, 0.5]: Jacobi-scaled cond kappa_s ~ 1e10 .. 1e11.  Kernel and band-solver oracle each sit inside
    eps * kappa_s of the exact solution (tests/test_force_truth.py measures both against a 50-digit solve: oracle
    <= 0.07, kernel <= 0.27 eps kappa_s), so they are within eps * kappa_s of each other -- displacements AND the end
    forces of `eleResponse(e,'forces')` (SingleCore.py:189-190), on every tiling incl. 64 lanes per beam."""
    g = load_golden(os.path.join(golden_dir, "bridge_adversarial.npz"))
    v, th, V, M, st = _solve(oa, g["x"], g["E"], g["I"], g["fix"], g["Fy"], g["wy"], tiling)
    assert (st == 0).all()
    for b in range(g["I"].shape[0]):
        tol = 2.2e-16 * kappa_scaled(g["x"], float(g["E"]), g["I"][b], g["fix"])
        assert tol < 1e-4
        for got, want in ((v, g["v"]), (th, g["theta"]), (V, g["V"]), (M, g["M"])):
            assert relerr(got[b], want[b]) < tol, (b, tol)


@pytest.mark.parametrize("B", [1, 3, 4, 5, 63, 64, 65, 2000])
def test_vs_c_oracle_seeded(oa, B):
    rng = np.random.default_rng(1000 + B)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, B, inertia="trajectory")
    ref = co.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF, n_threads=4)
    v, th, V, M, st = _solve(oa, x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
    assert (st == 0).all() and (ref[4] == 0).all()
    assert relerr(v, ref[0]) < 1e-8 and relerr(th, ref[1]) < 1e-8
    assert relerr(V, ref[2]) < 2e-6 and relerr(M, ref[3]) < 2e-6


@pytest.mark.parametrize("Ne", [1, 2, 3, 7, 13, 14, 50, 99, 103, 111, 127, 255, 511, 1023])
def test_ragged_sizes_nonuniform_mesh(oa, Ne):
    rng = np.random.default_rng(Ne)
    N = Ne + 1
    x = np.sort(rng.uniform(0, 3.0 * Ne, size=N)) + np.arange(N) * 0.5
    fix = np.zeros(N, dtype=np.uint8); fix[0] = 1; fix[-1] = 1
    if N > 4:
        fix[N // 3] = 1
    if Ne == 1:
        fix[0] = 3
    B = 5
    I = np.exp(rng.uniform(np.log(1e-2), np.log(0.5), size=(B, Ne)))
    Fy = rng.uniform(-1e5, 0, size=(B, N))
    E = np.full((B, Ne), 2.0e11) * rng.uniform(0.5, 1.5, size=(B, Ne))     # per-element E and wy paths
    wy = rng.uniform(-2000, 0, size=(B, Ne))
    ref = bo.solve_beam_batched(x, E, I, fix, Fy, wy)
    out = _solve(oa, x, E, I, fix, Fy, wy)
    K, _ = bo.assemble_beam(x, E[0], I[0], Fy[0], wy[0])
    free = np.ones(2 * N, dtype=bool); free[0::2] = (fix & 1) == 0; free[1::2] = (fix & 2) == 0
    tol = max(1e-10, 2e-16 * np.linalg.cond(K[np.ix_(free, free)]))
    assert (out[4] == 0).all()
    assert relerr(out[0], ref[0]) < tol and relerr(out[1], ref[1]) < tol
    # element end forces on every size, i.e. every tiling up to 64 lanes x 16 elements (interface refinement)
    assert relerr(out[2], ref[2]) < 10 * tol and relerr(out[3], ref[3]) < 10 * tol
    for P in sorted({p for p, m in TILINGS if p * m >= N and p not in FAT_P}):   # per-element E / wy: not the fat tiling's layout
        o2 = _solve(oa, x, E, I, fix, Fy, wy, tiling=P)
        assert relerr(o2[0], ref[0]) < tol and relerr(o2[2], ref[2]) < 10 * tol and relerr(o2[3], ref[3]) < 10 * tol, P


def test_clamped_rotation_fix_bit(oa):
    # OPS_AMD_FIX_RZ: cantilever PL^3/3EI
    N, L, EI, P = 41, 10.0, 2.0e7, -3.0e3
    x = np.linspace(0, L, N)
    fix = np.zeros(N, dtype=np.uint8); fix[0] = 3
    Fy = np.zeros((2, N)); Fy[:, -1] = P
    I = np.full((2, N - 1), EI / 2e11)
    v, th, V, M, st = _solve(oa, x, 2e11, I, fix, Fy, 0.0)
    assert v[0, -1] == pytest.approx(P * L**3 / (3 * EI), rel=1e-8)
    assert th[0, -1] == pytest.approx(P * L**2 / (2 * EI), rel=1e-8)
    assert M[0, 0] == pytest.approx(-P * L, rel=1e-8)


def test_status_flags_non_spd(oa):
    x = np.linspace(0, 10, 11)
    fix = np.zeros(11, dtype=np.uint8); fix[0] = fix[-1] = 1
    I = np.full((6, 10), 0.1); I[0, :] = 0.0; I[4, 4] = -0.1
    Fy = np.zeros((6, 11)); Fy[:, 5] = -1.0
    v, th, V, M, st = _solve(oa, x, 2e11, I, fix, Fy, 0.0)
    assert st[0] != 0 and st[4] != 0 and (st[[1, 2, 3, 5]] == 0).all()
    assert np.isnan(v[0]).all() and np.isnan(v[4]).all() and np.isfinite(v[[1, 2, 3, 5]]).all()


@pytest.mark.parametrize("tiling", [0, 8, 64])
def test_full_size_properties(oa, tiling):
    """BASELINE config 2 size (10 000 x 100): linearity, 1/I scaling, equilibrium, permutation invariance."""
    B = 10000
    rng = np.random.default_rng(20250307)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, B, inertia="trajectory")
    _, Fy2 = bo.random_cases(rng, B, inertia="uniform")
    xg, fg, Ig = _gpu(x), _gpu(fix, torch.uint8), _gpu(I)
    Eg, wg, zg = _gpu(bo.E_REF), _gpu(bo.UDL_REF), _gpu(0.0)
    F1, F2 = _gpu(Fy), _gpu(Fy2)
    a = oa.beam_solve(xg, Eg, Ig, fg, F1, wg, tiling=tiling)
    b = oa.beam_solve(xg, Eg, Ig, fg, F2, zg, tiling=tiling)
    c = oa.beam_solve(xg, Eg, Ig, fg, F1 + F2, wg, tiling=tiling)
    assert int(a.status.abs().sum()) == 0
    scale = a.v.abs().amax(dim=1, keepdim=True)
    assert float(((a.v + b.v - c.v).abs() / scale).max()) < 1e-9           # superposition
    assert float(((a.M + b.M - c.M).abs() / a.M.abs().amax(dim=1, keepdim=True)).max()) < 1e-6
    d = oa.beam_solve(xg, Eg, Ig * 4.0, fg, F1, wg, tiling=tiling)           # u ~ 1/I, forces unchanged
    assert float(((a.v - 4.0 * d.v).abs() / scale).max()) < 1e-9
    assert float(((a.V - d.V).abs() / a.V.abs().amax(dim=1, keepdim=True)).max()) < 1e-6
    perm = torch.randperm(B, device="cuda")
    e = oa.beam_solve(xg, Eg, Ig[perm].contiguous(), fg, F1[perm].contiguous(), wg, tiling=tiling)
    assert torch.equal(e.v, a.v[perm]) and torch.equal(e.M, a.M[perm])     # batch position must not matter
    # supports really are supports; global vertical equilibrium: sum of reactions = - applied load
    assert float(a.v[:, fg.bool()].abs().max()) == 0.0
    # shear jump across every free, unloaded node equals the UDL on one element (statics identity)
    V = a.V
    jump = V[:, 1:] - V[:, :-1]                     # nodal equilibrium: V_e - V_{e-1} = w L + F_node (free nodes)
    free_inner = (fg[1:-1] == 0)
    expect = bo.UDL_REF * 2.0 + F1[:, 1:-1]
    err = ((jump - expect)[:, free_inner]).abs().max() / V.abs().max()
    assert float(err) < 2e-6     # element end forces: same bound as the oracle comparison


def test_nan_loads_do_not_cross_beams(oa):
    """A beam whose loads are NaN must not contaminate its neighbours in the same wavefront (its row sits
    next to theirs in LDS): their results stay bit-identical, its own become NaN."""
    rng = np.random.default_rng(42)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, 12, inertia="trajectory")
    clean = _solve(oa, x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
    Fy2 = Fy.copy(); Fy2[5, :] = np.nan
    dirty = _solve(oa, x, bo.E_REF, I, fix, Fy2, bo.UDL_REF)
    keep = [b for b in range(12) if b != 5]
    for a, d in zip(clean[:4], dirty[:4]):
        assert np.array_equal(a[keep], d[keep])
    assert np.isnan(dirty[0][5]).all()


def test_strided_rows_through_the_c_abi(oa):
    """I and Fy rows with a batch stride larger than the row (the non-dense kernel variant), per-beam x / fix /
    E / wy at the same time: called through the C ABI directly, as a foreign host would."""
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    rng = np.random.default_rng(9)
    B, Ne = 37, 100
    N = Ne + 1
    g = load_golden(os.path.join(os.path.dirname(__file__), "golden", "random_bridge.npz"))
    reps = (B + g["I"].shape[0] - 1) // g["I"].shape[0]
    tile = lambda a: np.tile(a, (reps, 1))[:B]  # noqa: E731
    I, Fy, x, fix = tile(g["I"]), tile(g["Fy"]), tile(g["x"]), tile(g["fix"])
    E = np.full((B, Ne), float(g["E"])) * rng.uniform(0.8, 1.2, size=(B, Ne))
    wy = rng.uniform(-1500, -500, size=(B, Ne))
    ref = bo.solve_beam_batched(x, E, I, fix, Fy, wy)
    ks = [kappa_scaled(x[b], E[b], I[b], fix[b]) for b in range(B)]
    sI, sF = Ne + 3, N + 5
    Ipad = np.full((B, sI), np.nan); Ipad[:, :Ne] = I
    Fpad = np.full((B, sF), np.nan); Fpad[:, :N] = Fy
    d = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")  # noqa: E731
    dI, dF, dx, dfix, dE, dw = d(Ipad), d(Fpad), d(x), d(fix, torch.uint8), d(E), d(wy)
    out = [torch.empty((B, N), dtype=torch.float64, device="cuda") for _ in range(2)] + \
          [torch.empty((B, Ne), dtype=torch.float64, device="cuda") for _ in range(2)]
    st = torch.empty(B, dtype=torch.int32, device="cuda")
    for tiling in (0, 8, 64):
        rc = lib.ops_beam_solve_batched_f64(B, Ne, dx.data_ptr(), N, dE.data_ptr(), Ne, dI.data_ptr(), sI, dfix.data_ptr(), N,
                                            dF.data_ptr(), sF, dw.data_ptr(), Ne, out[0].data_ptr(), out[1].data_ptr(),
                                            out[2].data_ptr(), out[3].data_ptr(), st.data_ptr(), tiling,
                                            torch.cuda.current_stream().cuda_stream)
        assert rc == _cabi.OK
        torch.cuda.synchronize()
        assert int(st.abs().sum()) == 0
        # random bridges: cond(K) up to ~1e9 (one off-centre roller = a long soft cantilever); north_star's 1e-6
        assert relerr(out[0].cpu().numpy(), ref[0]) < 1e-6 and relerr(out[1].cpu().numpy(), ref[1]) < 1e-6
        # end forces: per beam, eps * kappa_s (Jacobi-scaled condition of that beam's stiffness matrix: what bounds the force error of
        # ANY Cholesky-type elimination, the band solver's included -- tests/test_force_truth.py measures both against 50 digits)
        Vg, Mg = out[2].cpu().numpy(), out[3].cpu().numpy()
        for b in range(B):
            ftol = max(1e-9, 50 * 2.2e-16 * ks[b])
            assert np.abs(Vg[b] - ref[2][b]).max() <= ftol * np.abs(ref[2][b]).max(), (tiling, b, ks[b])
            assert np.abs(Mg[b] - ref[3][b]).max() <= ftol * np.abs(ref[3][b]).max(), (tiling, b, ks[b])
    # bad strides are rejected, not dereferenced
    rc = lib.ops_beam_solve_batched_f64(B, Ne, dx.data_ptr(), N, dE.data_ptr(), Ne, dI.data_ptr(), Ne - 1, dfix.data_ptr(), N,
                                        dF.data_ptr(), sF, dw.data_ptr(), Ne, out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr(), out[3].data_ptr(), st.data_ptr(), 0, None)
    assert rc == _cabi.ERR_INVALID_ARG


def test_large_random_bridge_batch_vs_c_oracle(oa):
    """3 000 random bridges (per-beam length, 1-4 random rollers: SingleCore.py:133-151) against the C oracle."""
    from openpystruct_amd import sizing
    cases = sizing.make_cases(3000, sizing.SizingConfig(random_bridge=1), seed=5)
    rng = np.random.default_rng(6)
    I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=(3000, 100)))
    x, fix, Fy = cases.node_positions.numpy(), cases.fix.numpy(), cases.Fy.numpy()
    ref = co.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF, n_threads=4)
    v, th, V, M, st = _solve(oa, x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
    ok = ref[4] == 0
    assert (st[ok] == 0).all() and ok.mean() > 0.99
    # cond(K) varies by orders of magnitude with the support layout (a single off-centre roller leaves a long,
    # soft cantilever): compare where the oracle itself is trustworthy, relative to each beam's scale
    assert np.median(np.abs(v[ok] - ref[0][ok]).max(axis=1) / np.abs(ref[0][ok]).max(axis=1)) < 1e-7   # eps * cond, cond ~ 1e7-1e9
    assert relerr(v[ok], ref[0][ok]) < 1e-5


def test_torch_library_operator_matches_beam_solve_and_captures(oa):
    rng = np.random.default_rng(77)
    I, Fy = bo.random_cases(rng, 300, inertia="trajectory")
    args = (_gpu(np.linspace(0, 200, 101)), torch.tensor(bo.E_REF, dtype=torch.float64, device="cuda"), _gpu(I),
            _gpu(bo.reference_fix_mask(), torch.uint8), _gpu(Fy), torch.tensor(bo.UDL_REF, dtype=torch.float64, device="cuda"))
    ref = oa.beam_solve(*args)
    got = torch.ops.openpystruct_amd.beam_solve(*args)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    # inside a HIP graph (the operator allocates its outputs from the graph's private pool)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.ops.openpystruct_amd.beam_solve(*args)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = torch.ops.openpystruct_amd.beam_solve(*args)
    torch.cuda.current_stream().wait_stream(side)
    args[2].mul_(2.0)                                   # twice the inertia: half the deflection, same forces
    g.replay()
    torch.cuda.synchronize()
    assert float(relerr(out[0].cpu().numpy(), 0.5 * ref.v.cpu().numpy())) < 1e-9


@pytest.mark.parametrize("seed", range(12))
def test_random_support_patterns_sizes_and_tilings(oa, seed):
    """Random meshes (1..60 elements), random per-beam support patterns incl. fixed rotations and fully clamped nodes,
    random loads and section data, every tiling that fits: displacements vs the dense oracle to eps * cond."""
    rng = np.random.default_rng(9000 + seed)
    Ne = int(rng.integers(1, 61))
    N = Ne + 1
    B = 11
    x = np.cumsum(rng.uniform(0.2, 3.0, size=N))
    fix = np.zeros((B, N), dtype=np.uint8)
    for b in range(B):
        k = int(rng.integers(1, 5))
        nodes = rng.choice(N, size=min(k, N), replace=False)
        fix[b, nodes] = rng.integers(1, 4, size=nodes.size)                    # uy, rz or both
        if not (fix[b] & 1).any():
            fix[b, nodes[0]] |= 1                                              # some vertical support
        if ((fix[b] & 1).sum() < 2) and not ((fix[b] == 3).any()):
            fix[b, nodes[0]] = 3                                               # one vertical support only: clamp it
    I = np.exp(rng.uniform(np.log(1e-3), np.log(0.5), size=(B, Ne)))
    E = 2.0e11 * rng.uniform(0.5, 1.5, size=(B, Ne))
    Fy = rng.uniform(-1e5, 1e4, size=(B, N)) * (rng.random((B, N)) < 0.3)
    wy = rng.uniform(-2000, 0, size=(B, Ne))
    xb = np.tile(x, (B, 1))
    ref = [bo.solve_beam_dense(x, E[b], I[b], fix[b], Fy[b], wy[b]) for b in range(B)]
    conds = []
    for b in range(B):
        K, _ = bo.assemble_beam(x, E[b], I[b], Fy[b], wy[b])
        free = np.ones(2 * N, dtype=bool); free[0::2] = (fix[b] & 1) == 0; free[1::2] = (fix[b] & 2) == 0
        conds.append(np.linalg.cond(K[np.ix_(free, free)]) if free.any() else 1.0)
    for tiling in (0, 8, 16, 32, 64):
        if tiling and tiling * {8: 13, 16: 7, 32: 4, 64: 16}[tiling] < N:
            continue
        v, th, V, M, st = _solve(oa, xb, E, I, fix, Fy, wy, tiling=tiling)
        assert (st == 0).all()
        for b in range(B):
            tol = max(1e-9, 5e-16 * conds[b])
            sc = max(np.abs(ref[b][0]).max(), 1e-300)
            assert np.abs(v[b] - ref[b][0]).max() / sc < tol, (tiling, b, Ne)
            sct = max(np.abs(ref[b][1]).max(), 1e-300)
            assert np.abs(th[b] - ref[b][1]).max() / sct < tol, (tiling, b, Ne)
            assert (v[b][(fix[b] & 1) != 0] == 0.0).all() and (th[b][(fix[b] & 2) != 0] == 0.0).all()
            for got, want in ((V[b], ref[b][2]), (M[b], ref[b][3])):       # end forces, every tiling
                scf = max(np.abs(want).max(), 1e-300)
                assert np.abs(got - want).max() / scf < 10 * tol, (tiling, b, Ne)


@pytest.mark.parametrize("tiling", [0, 8, 16, 32, 64])
def test_reference_bridge_support_moments_by_the_three_moment_equation(oa, tiling):
    """The HIP solve of the reference's bridge under its UDL against Clapeyron's three-moment equation -- no oracle involved."""
    q = 1000.0
    xs = np.array([0.0, 18.0, 58.0, 138.0, 168.0, 198.0]); Ls = np.diff(xs)
    M5 = -q * 2.0 * 2.0 / 2.0
    A = np.zeros((4, 4)); rhs = np.zeros(4)
    for k, i in enumerate(range(1, 5)):
        A[k, k] = 2.0 * (Ls[i - 1] + Ls[i])
        if k > 0: A[k, k - 1] = Ls[i - 1]
        if k < 3: A[k, k + 1] = Ls[i]
        rhs[k] = -q * (Ls[i - 1] ** 3 + Ls[i] ** 3) / 4.0
    rhs[3] -= Ls[4] * M5
    Ms = np.concatenate([np.linalg.solve(A, rhs), [M5]])
    x = np.linspace(0.0, 200.0, 101)
    v, th, V, M, st = _solve(oa, x, bo.E_REF, np.full((9, 100), bo.I0_REF), bo.reference_fix_mask(), np.zeros((9, 101)), -q, tiling=tiling)
    assert (st == 0).all()
    for n, Mb in zip([10, 30, 70, 85, 100], Ms):
        assert -M[4, n - 1] == pytest.approx(Mb, rel=5e-8), (n, -M[4, n - 1], Mb)
    assert np.abs(v[:, [0, 9, 29, 69, 84, 99]]).max() == 0.0


@pytest.mark.parametrize("tiling", [0, 8, 32])
def test_mirror_symmetry(oa, tiling):
    """Solving the mirrored beam (x -> L - x, element and node arrays reversed) gives the mirrored field:
    v reversed, theta reversed with the opposite sign -- the lanes see completely different segments."""
    rng = np.random.default_rng(5150)
    B, Ne, N = 64, 100, 101
    x = np.cumsum(np.concatenate([[0.0], rng.uniform(0.5, 3.5, size=Ne)]))
    fix = np.zeros(N, dtype=np.uint8); fix[[3, 27, 55, 80, 97]] = 1; fix[12] = 3
    I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=(B, Ne)))
    Fy = rng.uniform(-3e5, 0.0, size=(B, N)) * (rng.random((B, N)) < 0.05)
    wy = rng.uniform(-2000.0, 0.0, size=(B, Ne))
    E = np.full((B, Ne), bo.E_REF)
    a = _solve(oa, x, E, I, fix, Fy, wy, tiling=tiling)
    xm = (x[-1] - x)[::-1].copy()
    b = _solve(oa, xm, E[:, ::-1].copy(), I[:, ::-1].copy(), fix[::-1].copy(), Fy[:, ::-1].copy(), wy[:, ::-1].copy(), tiling=tiling)
    assert (a[4] == 0).all() and (b[4] == 0).all()
    sv = np.abs(a[0]).max(axis=1, keepdims=True); st = np.abs(a[1]).max(axis=1, keepdims=True)
    assert (np.abs(b[0][:, ::-1] - a[0]) / sv).max() < 2e-8          # two different elimination orders: eps * cond apart
    assert (np.abs(b[1][:, ::-1] + a[1]) / st).max() < 2e-8


def test_stream_out_flag_gives_identical_results(oa):
    """OPS_AMD_TILING_STREAM_OUT only changes the cache policy of the result stores: bit-identical outputs, same kernel."""
    rng = np.random.default_rng(11)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, 37, inertia="trajectory")
    args = (_gpu(x), _gpu(bo.E_REF), _gpu(I), _gpu(fix, torch.uint8), _gpu(Fy), _gpu(bo.UDL_REF))
    for til in (0, 8, 16, 64):
        a = oa.beam_solve(*args, tiling=til)
        b = oa.beam_solve(*args, tiling=til, stream_out=True)
        torch.cuda.synchronize()
        for p, q in zip(a, b):
            assert torch.equal(p, q)
    from openpystruct_amd import _cabi
    assert _cabi.load().ops_beam_solve_kernel_name(10000, 100, 0x100) == _cabi.load().ops_beam_solve_kernel_name(10000, 100, 0)
