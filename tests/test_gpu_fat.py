"""GPU parity tests of the fat-wave tiling (csrc/beam_fat.hip: 6 lanes x 17 elements per beam, 10 beams per wave),
the kernel the default dispatch uses for the contract batch (10 000 beams x 100 elements).  Through the C ABI,
against the oracle on the same seeded inputs and against the 16-lane kernel; the golden-fixture, force-truth and
full-size property tests of test_gpu_parity.py / test_force_truth.py cover it too (it is in helpers.TILINGS).

Reference semantics under test: `setup_model` + `analyze(1)` + `eleResponse` + `nodeDisp`
(OpenPyStruct_BeamOpt_training_SingleCore.py:89-124, :180-190, :224-232)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import beam_oracle as bo  # noqa: E402
from tests.helpers import kappa_scaled, relerr  # noqa: E402

FAT = 6
ROWS = 0x200     # OPS_AMD_TILING_ROWS: the row-staged variants of the 16- and 8-lane tilings
VARIANTS = [FAT, 16 | ROWS, 8 | ROWS]


@pytest.fixture(scope="module")
def oa():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import openpystruct_amd as oa_
    from openpystruct_amd import _cabi

    _cabi.load()
    return oa_


def _gpu(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def _solve(oa, x, E, I, fix, Fy, wy, tiling=0):
    out = oa.beam_solve(_gpu(x), _gpu(E), _gpu(I), _gpu(fix, torch.uint8), _gpu(Fy), _gpu(wy), tiling=tiling)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out]


def test_kernel_names(oa):
    assert "beam_rows_kernel<6, 17, 1, true>" in oa.kernel_name(10000, 100, FAT)
    assert "beam_rows_kernel<16, 7, 3, false>" in oa.kernel_name(10000, 100, 16 | ROWS)
    assert "beam_rows_kernel<8, 13, 2, false>" in oa.kernel_name(10000, 100, 8 | ROWS)


@pytest.mark.parametrize("til", VARIANTS)
@pytest.mark.parametrize("B", [1, 9, 10, 11, 23, 640, 1003])
def test_ragged_batches_vs_oracle_and_16_lane_kernel(oa, B, til):
    """Batches that do not fill the last wave (10 beams per wave), the reference bridge with generator loads."""
    rng = np.random.default_rng(100 + B)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, B, inertia="trajectory")
    got = _solve(oa, x, bo.E_REF, I, fix, Fy, bo.UDL_REF, tiling=til)
    other = _solve(oa, x, bo.E_REF, I, fix, Fy, bo.UDL_REF, tiling=16)
    assert (got[4] == 0).all()
    nb = min(B, 64)
    ref = bo.solve_beam_batched(x, bo.E_REF, I[:nb], fix, Fy[:nb], bo.UDL_REF)
    for k, tol in ((0, 1e-8), (1, 1e-8), (2, 2e-6), (3, 2e-6)):
        assert relerr(got[k][:nb], ref[k]) < tol
        assert relerr(got[k], other[k]) < tol          # every beam of the batch against the 16-lane kernel


@pytest.mark.parametrize("til", VARIANTS)
@pytest.mark.parametrize("Ne", [1, 2, 5, 16, 17, 18, 50, 84, 85, 86, 99, 100, 101])
def test_element_counts_odd_and_even(oa, Ne, til):
    """Every kind of row the staging meets: odd / even element and node counts (16-byte pairs + a tail double), lanes
    that own padding only (Ne <= 85: the sixth lane of a beam holds nothing real), the largest size the tiling serves."""
    rng = np.random.default_rng(Ne)
    N = Ne + 1
    B = 13
    x = np.sort(rng.uniform(0, 3.0 * Ne, size=N)) + np.arange(N) * 0.5
    fix = np.zeros(N, dtype=np.uint8); fix[0] = 1; fix[-1] = 1
    if N > 4:
        fix[N // 3] = 1
    if Ne == 1:
        fix[0] = 3
    I = np.exp(rng.uniform(np.log(1e-2), np.log(0.5), size=(B, Ne)))
    Fy = rng.uniform(-1e5, 0, size=(B, N))
    ref = bo.solve_beam_batched(x, 2.0e11, I, fix, Fy, -750.0)
    got = _solve(oa, x, 2.0e11, I, fix, Fy, -750.0, tiling=til)
    assert (got[4] == 0).all()
    K, _ = bo.assemble_beam(x, 2.0e11, I[0], Fy[0], -750.0)
    free = np.ones(2 * N, dtype=bool); free[0::2] = (fix & 1) == 0; free[1::2] = (fix & 2) == 0
    tol = max(3e-10, 4e-16 * np.linalg.cond(K[np.ix_(free, free)]))     # two elimination orders, each at ~1e-16 * cond
    assert relerr(got[0], ref[0]) < tol and relerr(got[1], ref[1]) < tol
    ks = max(kappa_scaled(x, 2.0e11, I[0], fix), 1.0)
    ftol = max(1e-9, 50 * 2.2e-16 * ks)
    assert relerr(got[2], ref[2]) < ftol and relerr(got[3], ref[3]) < ftol


@pytest.mark.parametrize("til", VARIANTS)
def test_fixed_rotations_take_the_general_path(oa, til):
    """Clamped supports (`fix(n, ., 1, 1)`): the wave-uniform branch with the rotation flags."""
    rng = np.random.default_rng(3)
    x = np.linspace(0, 60, 101)
    fix = np.zeros(101, dtype=np.uint8); fix[0] = 3; fix[40] = 1; fix[70] = 2; fix[100] = 3
    I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=(25, 100)))
    Fy = rng.uniform(-3e5, 0.0, size=(25, 101)) * (rng.random((25, 101)) < 0.1)
    ref = bo.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
    got = _solve(oa, x, bo.E_REF, I, fix, Fy, bo.UDL_REF, tiling=til)
    assert (got[4] == 0).all()
    for k, tol in ((0, 1e-8), (1, 1e-8), (2, 2e-6), (3, 2e-6)):
        assert relerr(got[k], ref[k]) < tol
    assert np.abs(got[0][:, [0, 40, 100]]).max() == 0.0 and np.abs(got[1][:, [0, 70, 100]]).max() == 0.0   # constrained DOFs: exactly 0


@pytest.mark.parametrize("til", VARIANTS)
def test_bad_beams_are_flagged_and_do_not_cross(oa, til):
    """A non-positive pivot (what makes dpbsv / `analyze` fail, MultiCore.py:182-186) flags that beam only; NaN inputs
    of one beam leave its neighbours in the wave bit-identical."""
    rng = np.random.default_rng(42)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, 27, inertia="trajectory")
    clean = _solve(oa, x, bo.E_REF, I, fix, Fy, bo.UDL_REF, tiling=til)
    I2, Fy2 = I.copy(), Fy.copy()
    I2[3, :] = 0.0; I2[14, 40] = -0.1; Fy2[5, :] = np.nan; I2[19, 99] = np.nan; I2[26, 0] = np.inf
    dirty = _solve(oa, x, bo.E_REF, I2, fix, Fy2, bo.UDL_REF, tiling=til)
    hit = [3, 5, 14, 19, 26]
    keep = [b for b in range(27) if b not in hit]
    for a, d in zip(clean[:4], dirty[:4]):
        assert np.array_equal(a[keep], d[keep])
    assert (dirty[4][keep] == 0).all() and (dirty[4][[3, 14]] != 0).all()
    for b in (3, 14, 5, 19):
        assert np.isnan(dirty[0][b]).all() and np.isnan(dirty[2][b]).all()


@pytest.mark.parametrize("til", VARIANTS)
def test_strided_misaligned_rows_and_forces_only_through_the_c_abi(oa, til):
    """Row strides larger than the rows, row bases that are only 8-byte aligned, and the forces-only entry point."""
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    rng = np.random.default_rng(9)
    B, Ne = 37, 100
    N = Ne + 1
    x = np.linspace(0, 200, N)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, B, inertia="trajectory")
    ref = bo.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
    sI, sF = Ne + 3, N + 4
    big = torch.full((B * sI + 1,), float("nan"), dtype=torch.float64, device="cuda")
    dI = big[1:].view(B, sI)                      # base 8 bytes off a 16-byte boundary
    dI[:, :Ne] = _gpu(I)
    bigF = torch.full((B * sF + 1,), float("nan"), dtype=torch.float64, device="cuda")
    dF = bigF[1:].view(B, sF)
    dF[:, :N] = _gpu(Fy)
    dx, dfix = _gpu(x), _gpu(fix, torch.uint8)
    dE, dw = _gpu(bo.E_REF), _gpu(bo.UDL_REF)
    obuf = [torch.full((B * n + 1,), -7.0, dtype=torch.float64, device="cuda") for n in (N, N, Ne, Ne)]
    out = [o[1:].view(B, -1) for o in obuf]
    st = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.ops_beam_solve_batched_f64(B, Ne, dx.data_ptr(), 0, dE.data_ptr(), 0, dI.data_ptr(), sI, dfix.data_ptr(), 0,
                                        dF.data_ptr(), sF, dw.data_ptr(), 0, out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr(), out[3].data_ptr(), st.data_ptr(), til, stream)
    assert rc == _cabi.OK
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    for k, tol in ((0, 1e-8), (1, 1e-8), (2, 2e-6), (3, 2e-6)):
        assert relerr(out[k].cpu().numpy(), ref[k]) < tol
    assert all(float(o[0]) == -7.0 for o in obuf)            # nothing written in front of the rows
    V2 = torch.zeros((B, Ne), dtype=torch.float64, device="cuda")
    M2 = torch.zeros_like(V2)
    rc = lib.ops_beam_solve_forces_f64(B, Ne, dx.data_ptr(), 0, dE.data_ptr(), 0, dI.data_ptr(), sI, dfix.data_ptr(), 0,
                                       dF.data_ptr(), sF, dw.data_ptr(), 0, V2.data_ptr(), M2.data_ptr(), st.data_ptr(), None, til, stream)
    assert rc == _cabi.OK
    torch.cuda.synchronize()
    assert torch.equal(V2, out[2]) and torch.equal(M2, out[3])
    # per-beam geometry or masks are not this tiling's: an explicit request says so instead of silently using another kernel
    dxb = _gpu(np.tile(x, (B, 1)))
    rc = lib.ops_beam_solve_batched_f64(B, Ne, dxb.data_ptr(), N, dE.data_ptr(), 0, dI.data_ptr(), sI, dfix.data_ptr(), 0,
                                        dF.data_ptr(), sF, dw.data_ptr(), 0, out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr(), out[3].data_ptr(), st.data_ptr(), til, stream)
    assert rc == _cabi.ERR_UNSUPPORTED


@pytest.mark.parametrize("til", VARIANTS)
def test_stream_out_is_bit_identical_at_the_contract_batch(oa, til):
    rng = np.random.default_rng(11)
    x = np.linspace(0, 200, 101)
    I, Fy = bo.random_cases(rng, 10000, inertia="trajectory")
    args = (_gpu(x), _gpu(bo.E_REF), _gpu(I), _gpu(bo.reference_fix_mask(), torch.uint8), _gpu(Fy), _gpu(bo.UDL_REF))
    a = oa.beam_solve(*args, tiling=til)
    b = oa.beam_solve(*args, tiling=til, stream_out=True)
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    d = oa.beam_solve(*args, tiling=16)
    torch.cuda.synchronize()
    for k, tol in ((0, 1e-8), (1, 1e-8), (2, 2e-6), (3, 2e-6)):
        assert relerr(a[k].cpu().numpy(), d[k].cpu().numpy()) < tol
    assert int(a.status.abs().sum()) == 0
