"""Batched 2-D frame solve (HIP, LDS-resident band LDL^T) against the OpenSees-like 3-DOF oracle
(oracle/beam_oracle.py::solve_model_3dof = LAPACK dpbsv), incl. the reference's Wx = Wy quirk."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import beam_oracle as bo  # noqa: E402
from tests.helpers import relerr  # noqa: E402


@pytest.fixture(autouse=True)
def _tuned_kernels_for_every_batch():
    """The tests of this file launch 2-11 frames and are about the tuned kernels (packed / wave-per-frame); since r05 a batch of at most 256
    frames takes the workgroup-per-frame kernels (latency: csrc/frame_solve.hip latency_batch).  Library option frame_latency_batch = 0: tuned
    kernels for every batch; the tests that take `dispatch` run both ways.  Options are process-wide: put back after each test."""
    from openpystruct_amd import _cabi
    _cabi.set_option("frame_latency_batch", 0)
    yield
    _cabi.set_option("frame_latency_batch", -1)
    _cabi.set_option("frame_pack", 1)
    _cabi.set_option("frame_coop", 1)


@pytest.fixture(params=["wave", "latency"])
def dispatch(request):
    if request.param == "latency":
        from openpystruct_amd import _cabi
        _cabi.set_option("frame_latency_batch", -1)      # the library's default: B <= 256 -> a workgroup per frame
    return request.param


def _oracle(topo, I):
    return bo.solve_model_3dof(topo.coords, topo.conn, topo.A, topo.E, I, topo.fix3, topo.nodal_loads, wy=topo.wy, wx=topo.wx)


def _kd_ok(topo, oracle_kd):
    """The oracle numbers equations in node order; so does the product unless reverse Cuthill-McKee found a narrower band (r05)."""
    return topo.kd == oracle_kd if topo.numbering == "node" else topo.kd < oracle_kd


@pytest.mark.parametrize("bays,stories", [(1, 1), (2, 3), (4, 2), (7, 5), (10, 10)])
def test_grid_frames_vs_oracle(dispatch, bays, stories):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from openpystruct_amd import frames
    topo = frames.grid_frame(bays, stories)
    assert topo.Ne == stories * (bays + 1) + stories * bays and topo.Nn == (stories + 1) * (bays + 1)   # FR:66-69
    rng = np.random.default_rng(bays * 100 + stories)
    B = 5
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    torch.cuda.synchronize()
    assert int(sol.status.abs().sum()) == 0
    for b in range(B):
        d, f, st, neq, kd = _oracle(topo, I[b])
        assert st == 0 and neq == topo.n_eq and _kd_ok(topo, kd)
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7
        np.testing.assert_array_equal(sol.V[b].cpu().numpy(), sol.forces[b, :, 1].cpu().numpy())
        np.testing.assert_array_equal(sol.M[b].cpu().numpy(), sol.forces[b, :, 2].cpu().numpy())


def test_frame_equilibrium_and_axial_udl_quirk(dispatch):
    from openpystruct_amd import frames
    cfg = frames.FrameConfig()
    topo = frames.grid_frame(3, 2, cfg)
    I = torch.full((1, topo.Ne), cfg.I0, dtype=torch.float64, device="cuda")
    sol = frames.frame_solve(topo, I)
    f = sol.forces[0].cpu().numpy()
    # global equilibrium: the column bases (elements 0..bays of story 0) carry all applied load
    base = f[: 4]
    applied_x = cfg.lateral_load * 2 + 0.0
    applied_y = cfg.vertical_load * cfg.bay_width * 3 * 2          # Wy on 6 beams
    # beamUniform(w, w): the SAME w also acts along the beam axis (global x for beams): 6 beams * w * L
    applied_x += cfg.vertical_load * cfg.bay_width * 3 * 2
    assert base[:, 0].sum() == pytest.approx(-applied_x, rel=1e-9)
    assert base[:, 1].sum() == pytest.approx(-applied_y, rel=1e-9)


def test_frame_status_and_half_bandwidths_beyond_the_wave(dispatch):
    from openpystruct_amd import frames
    topo = frames.grid_frame(2, 2)
    I = torch.full((3, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
    I[1, 3] = -1.0
    sol = frames.frame_solve(topo, I)
    st = sol.status.cpu().numpy()
    assert st[1] != 0 and st[0] == 0 and st[2] == 0 and torch.isnan(sol.disp[1]).all() and torch.isfinite(sol.disp[0]).all()
    # half bandwidths beyond 63 (r05: the column-by-column fallback, csrc/frame_solve.hip frame_wide_kernel; up to r04 these raised)
    from oracle import beam_oracle as bo
    rng = np.random.default_rng(63)
    assert frames.grid_frame(21, 3).kd == 11               # (numberer('RCM'), FR:135: 11 along the column lines of what is 68 story by story)
    for topo in (frames.grid_frame(21, 3, numbering="node"),            # story by story: half bandwidth 3 * 22 + 2 = 68
                 frames.grid_frame(21, 21, numbering="auto"),           # no numbering helps a square 21 x 21 grid below 63
                 frames.grid_frame(30, 2, numbering="node")):           # 95
        assert topo.kd > 63
        I = np.exp(rng.uniform(np.log(1e-4), np.log(5e-3), size=(3, topo.Ne)))
        I[1, 5] = -1.0                                                   # not positive definite: status, NaN rows
        sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
        st = sol.status.cpu().numpy()
        assert st[0] == 0 and st[2] == 0 and st[1] != 0 and torch.isnan(sol.disp[1]).all()
        for k in (0, 2):
            d, f, s_, _, _ = bo.solve_model_3dof(topo.coords, topo.conn, topo.A, topo.E, I[k], topo.fix3, topo.nodal_loads, wy=topo.wy, wx=topo.wx)
            assert s_ == 0
            assert np.abs(sol.disp[k].cpu().numpy() - d).max() <= 1e-7 * np.abs(d).max(), (topo.kd, k)
            assert np.abs(sol.forces[k].cpu().numpy() - f).max() <= 1e-6 * np.abs(f).max(), (topo.kd, k)


@pytest.mark.parametrize("bays,stories,kd_node,kd_rcm", [(10, 2, 35, 8), (16, 3, 53, 11), (21, 3, 68, 11), (9, 4, 32, 14), (10, 10, 35, 32)])
def test_reverse_cuthill_mckee_numbering_vs_oracle(bays, stories, kd_node, kd_rcm):
    """r05: `FrameTopology(numbering="auto")` numbers the equations by reverse Cuthill-McKee when that narrows the band (the
    reference asks OpenSees for `numberer('RCM')`, FR:135): wide, low frames of the script's own random range (bays, stories ~ U{1..10})
    are banded along their column lines.  Same answers as the node-order oracle (and as the node-order product path where that exists),
    a fraction of the work."""
    from openpystruct_amd import frames
    topo = frames.grid_frame(bays, stories)
    assert topo.numbering != "node" and topo.kd == kd_rcm
    assert frames.grid_frame(bays, stories, numbering="rcm").kd <= kd_node
    eq = topo.d_node_eq.cpu().numpy()
    free = eq[eq >= 0]
    assert sorted(free.tolist()) == list(range(topo.n_eq))                   # a permutation of the equations
    rng = np.random.default_rng(bays + stories)
    B = 5
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    It = torch.as_tensor(I, device="cuda")
    sol = frames.frame_solve(topo, It)
    assert int(sol.status.abs().sum()) == 0
    for b in range(B):
        d, f, st, neq, okd = _oracle(topo, I[b])
        assert st == 0 and neq == topo.n_eq and okd == kd_node
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7
    if kd_node <= 63:
        ref = frames.frame_solve(frames.grid_frame(bays, stories, numbering="node"), It)
        assert relerr(sol.disp.cpu().numpy().ravel(), ref.disp.cpu().numpy().ravel()) < 1e-9
    # the sizing loop on it: same early stop and design as on the node-order numbering (float32 optimiser: to rounding)
    if kd_node <= 63:
        cfg = frames.FrameConfig()
        Ia, _, epa = frames.optimize_frames(topo, 2, cfg, max_epochs=60, poll_every=10)
        Ib, _, epb = frames.optimize_frames(frames.grid_frame(bays, stories, numbering="node"), 2, cfg, max_epochs=60, poll_every=10)
        assert torch.equal(epa, epb) and float((Ia - Ib).abs().max() / Ib.abs().max()) < 1e-4


@pytest.mark.parametrize("bays,stories", [(15, 16), (13, 13), (18, 20)])
def test_large_frames_band_in_hbm_workspace(dispatch, bays, stories):
    """BASELINE config 5 size (15 x 16 = 496 elements): the band no longer fits LDS and streams through a window."""
    from openpystruct_amd import _cabi, frames
    topo = frames.grid_frame(bays, stories)
    B = 3
    assert int(_cabi.load().ops_frame_workspace_bytes(B, topo.n_eq, topo.kd)) >= B * topo.lds_bytes() > 0   # one band + rhs per frame
    rng = np.random.default_rng(bays)
    I = np.exp(rng.uniform(np.log(1e-4), np.log(5e-3), size=(B, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    torch.cuda.synchronize()
    assert int(sol.status.abs().sum()) == 0
    for b in range(B):
        d, f, st, neq, kd = _oracle(topo, I[b])
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-7
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-6
    # second call re-uses the workspace; assembly uses float atomics, so equality is to rounding, not bitwise
    sol2 = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    assert relerr(sol2.disp.cpu().numpy().reshape(B, -1), sol.disp.cpu().numpy().reshape(B, -1)) < 1e-9


def test_frame_sizing_loop_runs_like_the_reference():
    """FR:163-206 batched: every frame early-stops, inertias stay positive, loss decreases."""
    from openpystruct_amd import frames
    topo = frames.grid_frame(3, 3)
    I, sol, ep = frames.optimize_frames(topo, 8, max_epochs=400)
    assert (ep.cpu().numpy() > 5).all() and float(I.min()) >= 1e-8 and int(sol.status.abs().sum()) == 0
    assert torch.allclose(I[0], I[7])          # identical problems -> identical trajectories, whatever the batch slot


def test_ops_shim_runs_setup_frame_model(dispatch):
    """The reference's frame script drives the same command API (FR:75-139, :151, :181-183): re-typed here."""
    from openpystruct_amd import frames, ops
    cfg = frames.FrameConfig()
    bays, stories = 3, 2
    nb1 = bays + 1
    coords = {i * nb1 + j + 1: (j * cfg.bay_width, i * cfg.story_height) for i in range(stories + 1) for j in range(nb1)}
    n_cols, n_beams = stories * nb1, stories * bays
    I = np.full(n_cols + n_beams, cfg.I0) * np.linspace(0.5, 2.0, n_cols + n_beams)
    ops.wipe()
    ops.model('basic', '-ndm', 2, '-ndf', 3)
    ops.geomTransf('Linear', 1)
    for tag, (x, y) in coords.items():
        ops.node(tag, x, y)
    for tag, (x, y) in coords.items():
        if y == 0.0:
            ops.fix(tag, 1, 1, 1)
    e = 1
    for i in range(stories):
        for j in range(nb1):
            ops.element('elasticBeamColumn', e, i * nb1 + j + 1, (i + 1) * nb1 + j + 1, cfg.A, cfg.E, float(I[e - 1]), 1); e += 1
    for i in range(1, stories + 1):
        for j in range(bays):
            ops.element('elasticBeamColumn', e, i * nb1 + j + 1, i * nb1 + j + 2, cfg.A, cfg.E, float(I[e - 1]), 1); e += 1
    ops.timeSeries('Linear', 1); ops.pattern('Plain', 1, 1)
    for tag, (x, y) in coords.items():
        if x == 0.0 and y != 0.0:
            ops.load(tag, cfg.lateral_load, 0.0, 0.0)
    for ele in range(n_cols + 1, n_cols + n_beams + 1):
        ops.eleLoad('-ele', ele, '-type', '-beamUniform', cfg.vertical_load, cfg.vertical_load)
    ops.system('BandGeneral'); ops.numberer('RCM'); ops.constraints('Plain'); ops.integrator('LoadControl', 1.0)
    ops.algorithm('Newton'); ops.analysis('Static')
    assert ops.analyze(1) == 0
    topo = frames.grid_frame(bays, stories, cfg, device="cpu")
    d, f, st, _, _ = _oracle(topo, I)
    for ele in (1, n_cols, n_cols + 1, n_cols + n_beams):
        np.testing.assert_allclose(ops.eleResponse(ele, 'forces'), f[ele - 1], rtol=1e-7, atol=1e-6 * np.abs(f).max())
    for n in (5, 8, 12):
        for dof in (1, 2, 3):
            assert ops.nodeDisp(n, dof) == pytest.approx(d[n - 1, dof - 1], rel=1e-7, abs=1e-12)


def _custom_frame(bays, stories, pinned, brace):
    """Grid frame with pinned (rotation-free) bases and optional diagonal braces: the number of equations is not a
    multiple of three and elements are inclined, so the padding equations and the rotation terms are exercised."""
    from openpystruct_amd import frames
    nb1 = bays + 1
    coords = np.array([(j * 4.0, i * 3.0) for i in range(stories + 1) for j in range(nb1)])
    conn = [(i * nb1 + j, (i + 1) * nb1 + j) for i in range(stories) for j in range(nb1)]
    conn += [(i * nb1 + j, i * nb1 + j + 1) for i in range(1, stories + 1) for j in range(bays)]
    if brace:
        conn += [(i * nb1, (i + 1) * nb1 + 1) for i in range(stories)]
    conn = np.array(conn)
    fix3 = np.zeros((coords.shape[0], 3), dtype=bool)
    fix3[coords[:, 1] == 0.0] = (True, True, not pinned)
    if pinned and (int((~fix3).sum()) % 3) == 0:
        fix3[0] = True                                     # clamp one base: keeps the count off a multiple of three
    loads = np.zeros((coords.shape[0], 3))
    loads[(coords[:, 0] == 0.0) & (coords[:, 1] != 0.0), 0] = 2.5e4
    loads[-1] = (0.0, -4e4, 1e3)
    w = np.zeros(len(conn)); w[stories * nb1: stories * nb1 + stories * bays] = -1.2e4
    return frames.FrameTopology(coords, conn, fix3, 0.02, 200e9, w, 0.5 * w, loads, "cuda")


@pytest.mark.parametrize("bays,stories,pinned,brace", [(1, 1, True, False), (2, 2, True, True), (3, 4, True, True),
                                                        (6, 5, True, False), (9, 9, True, True), (1, 5, False, True)])
def test_general_topologies_vs_oracle(dispatch, bays, stories, pinned, brace):
    from openpystruct_amd import frames
    topo = _custom_frame(bays, stories, pinned, brace)
    if pinned:
        assert topo.n_eq % 3 != 0
    rng = np.random.default_rng(7 * bays + stories)
    B = 4
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    assert int(sol.status.abs().sum()) == 0
    for b in range(B):
        d, f, st, neq, kd = _oracle(topo, I[b])
        assert st == 0 and neq == topo.n_eq and _kd_ok(topo, kd)
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7


def test_hub_node_with_eighteen_elements_takes_the_extra_plan_blocks(dispatch):
    """A node with eighteen incident elements: its row group holds more than the 192 assembly entries of one plan block
    (csrc/frame_wave.hpp FW_EPG), so the solve walks the group's extra blocks -- the path the grid frames never take."""
    from openpystruct_amd import frames
    nn, hub = 19, 9
    ang = np.linspace(0.0, 2 * np.pi, nn - 1, endpoint=False)
    coords = np.zeros((nn, 2))
    outer = [i for i in range(nn) if i != hub]
    coords[outer, 0], coords[outer, 1] = 5.0 * np.cos(ang), 5.0 * np.sin(ang)
    conn = [(hub, o) for o in outer] + [(outer[i], outer[i + 1]) for i in range(len(outer) - 1)]
    conn = np.array(conn)
    fix3 = np.zeros((nn, 3), dtype=bool)
    fix3[0] = fix3[nn - 1] = True
    loads = np.zeros((nn, 3)); loads[hub] = (3e4, -5e4, 2e3); loads[3] = (0.0, -1e4, 0.0)
    w = np.zeros(len(conn)); w[:4] = -8e3
    topo = frames.FrameTopology(coords, conn, fix3, 0.02, 200e9, w, 0.5 * w, loads, "cuda", numbering="node")   # the hub in the middle of the numbering
    eq = topo.d_elem_eq.cpu().numpy()
    per_group = np.zeros(topo.n_eq // 8 + 1, dtype=int)
    for e in range(topo.Ne):
        for r in range(6):
            if eq[e, r] >= 0:
                per_group[eq[e, r] // 8] += int(((eq[e] >= 0) & (eq[e] <= eq[e, r])).sum())
    assert per_group.max() > 192 and topo.kd <= 55
    rng = np.random.default_rng(11)
    B = 6
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    assert int(sol.status.abs().sum()) == 0
    for b in range(B):
        d, f, st, neq, kd = _oracle(topo, I[b])
        assert st == 0 and neq == topo.n_eq and _kd_ok(topo, kd)
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7


@pytest.mark.parametrize("skip,kd", [(2, 8), (4, 14), (7, 23), (10, 32), (12, 38), (16, 50), (17, 53)])
def test_ladder_frames_cover_every_window_width(dispatch, skip, kd):
    """Chains whose node i is also tied to node i + skip: half bandwidth 3 skip + 2, so every compiled register-window
    width of the wave kernel (16, 24, 36, 52, 56) and the multiples of eight (a row group enters exactly when the window
    reaches it) meet the oracle on a frame whose equation count is not a multiple of the group size."""
    from openpystruct_amd import frames
    nn = 3 * skip + 11
    coords = np.array([(1.5 * i, 0.4 * np.sin(0.9 * i) + 0.05 * (i % 3)) for i in range(nn)])
    conn = np.array([(i, i + 1) for i in range(nn - 1)] + [(i, i + skip) for i in range(nn - skip)])
    fix3 = np.zeros((nn, 3), dtype=bool)
    fix3[0] = True
    fix3[nn - 1, :2] = True
    loads = np.zeros((nn, 3)); loads[nn // 2] = (1e4, -3e4, 5e2); loads[nn // 3, 1] = -2e4
    w = np.zeros(len(conn)); w[: nn - 1] = -5e3
    topo = frames.FrameTopology(coords, conn, fix3, 0.02, 200e9, w, 0.5 * w, loads, "cuda", numbering="node")     # the window width under test
    assert topo.kd == kd and topo.n_eq == 3 * nn - 5
    rng = np.random.default_rng(skip)
    B = 5
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    assert int(sol.status.abs().sum()) == 0
    for b in range(B):
        d, f, st, neq, okd = _oracle(topo, I[b])
        assert st == 0 and neq == topo.n_eq and okd == topo.kd == kd
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7


def test_tiny_bandwidth_cantilever_chain(dispatch):
    """A single cantilever of 40 collinear elements: half bandwidth 5; and a two-node model: half bandwidth 2 < 3."""
    from openpystruct_amd import frames
    for nel in (40, 1):
        coords = np.array([(0.7 * i, 0.0) for i in range(nel + 1)])
        conn = np.array([(i, i + 1) for i in range(nel)])
        fix3 = np.zeros((nel + 1, 3), dtype=bool); fix3[0] = True
        loads = np.zeros((nel + 1, 3)); loads[-1] = (1e3, -2e3, 50.0)
        topo = frames.FrameTopology(coords, conn, fix3, 0.01, 200e9, -500.0, 0.0, loads, "cuda")
        I = np.full((2, nel), 3e-4)
        sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
        d, f, st, neq, kd = _oracle(topo, I[0])
        assert int(sol.status.abs().sum()) == 0 and _kd_ok(topo, kd)
        assert relerr(sol.disp[0].cpu().numpy().ravel(), d.ravel()) < 1e-8
        assert relerr(sol.forces[1].cpu().numpy().ravel(), f.ravel()) < 1e-7


def test_l_shaped_cantilever_closed_form_on_the_gpu(dispatch):
    """Statically determinate L (clamped column + arm, vertical tip load): the HIP frame solve against the closed form."""
    from openpystruct_amd import frames
    h, a, P = 4.0, 3.0, -2.0e4
    E, A, I = 2.0e11, 8.0e-3, 3.0e-5
    nc, na = 8, 6
    coords = np.array([(0.0, h * i / nc) for i in range(nc + 1)] + [(a * j / na, h) for j in range(1, na + 1)])
    conn = np.array([(i, i + 1) for i in range(nc + na)])
    fix3 = np.zeros((len(coords), 3), dtype=bool); fix3[0] = True
    loads = np.zeros((len(coords), 3)); loads[-1, 1] = P
    topo = frames.FrameTopology(coords, conn, fix3, A, E, 0.0, 0.0, loads, "cuda")
    sol = frames.frame_solve(topo, torch.full((3, len(conn)), I, dtype=torch.float64, device="cuda"))
    d = sol.disp[1].cpu().numpy(); f = sol.forces[2].cpu().numpy()
    EI, EA = E * I, E * A
    assert d[nc, 1] == pytest.approx(P * h / EA, rel=1e-9) and d[nc, 2] == pytest.approx(P * a * h / EI, rel=1e-9)
    assert d[nc, 0] == pytest.approx(-P * a * h * h / (2 * EI), rel=1e-9)
    assert d[-1, 1] == pytest.approx(P * h / EA + d[nc, 2] * a + P * a ** 3 / (3 * EI), rel=1e-9)
    assert f[0, 1] == pytest.approx(-P, rel=1e-9) and f[0, 2] == pytest.approx(-P * a, rel=1e-9)


@pytest.mark.parametrize("bays,stories", [(2, 2), (3, 2)])
def test_frame_sizing_loop_vs_per_frame_oracle(dispatch, bays, stories):
    """`frames.optimize_frames` against the per-frame restatement of FR:141-206 (oracle/frame_sizing_oracle.py: the
    reference's own torch calls around the 3-DOF band solve): inertias after each of the first epochs to float32 rounding
    (rtol 2e-5: float32 accumulation order differs -- the script adds element by element, the kernel reduces in a tree),
    early-stop epoch within +-3, final design within 1e-3."""
    from openpystruct_amd import frames
    from oracle import frame_sizing_oracle as fo
    cfg = frames.FrameConfig()
    topo = frames.grid_frame(bays, stories, cfg)
    rng = np.random.default_rng(bays * 10 + stories)
    B = 3
    I0 = np.full((B, topo.Ne), cfg.I0)
    I0[1] *= rng.uniform(0.5, 2.0, size=topo.Ne)
    I0[2] *= rng.uniform(0.2, 4.0, size=topo.Ne)
    I0 = I0.astype(np.float32)
    ref = [fo.optimize_frame(topo.coords, topo.conn, topo.fix3, topo.nodal_loads, topo.wy, topo.wx, A=cfg.A, E=cfg.E, nu=cfg.nu,
                             I0=cfg.I0, alpha_moment=cfg.alpha_moment, alpha_shear=cfg.alpha_shear, k=cfg.k, num_epochs=cfg.num_epochs,
                             lr=cfg.lr, tolerance=cfg.tolerance, patience=cfg.patience, I_init=I0[b]) for b in range(B)]
    for n in (1, 2, 3, 10):
        I, sol, ep = frames.optimize_frames(topo, B, cfg, I0=torch.as_tensor(I0, device="cuda"), max_epochs=n, poll_every=1)
        got = I.cpu().numpy()
        for b in range(B):
            assert ref[b]["epochs_run"] > n
            np.testing.assert_allclose(got[b], ref[b]["I_history"][n - 1], rtol=2e-5 * n, atol=0)
    I, sol, ep = frames.optimize_frames(topo, B, cfg, I0=torch.as_tensor(I0, device="cuda"))
    ep = ep.cpu().numpy()
    for b in range(B):
        assert abs(int(ep[b]) - ref[b]["epochs_run"]) <= 3, (b, ep[b], ref[b]["epochs_run"])
        # the last solve's forces: the state BEFORE the last step (one-step lag, as in the beam scripts)
        np.testing.assert_allclose(I[b].cpu().numpy(), ref[b]["I"], rtol=2e-3)


def test_frame_sizing_first_epochs_on_the_largest_frame_the_reference_draws(dispatch):
    """10 x 10 bays x stories (FR:17-18: the upper end of the script's random range; 210 elements, 330 equations): the first five
    epochs of `optimize_frames` against the per-frame restatement of FR:141-206, inertias to float32 rounding."""
    from openpystruct_amd import frames
    from oracle import frame_sizing_oracle as fo
    cfg = frames.FrameConfig()
    topo = frames.grid_frame(10, 10, cfg)
    rng = np.random.default_rng(1010)
    B = 2
    I0 = np.full((B, topo.Ne), cfg.I0)
    I0[1] *= rng.uniform(0.5, 2.0, size=topo.Ne)
    I0 = I0.astype(np.float32)
    ref = [fo.optimize_frame(topo.coords, topo.conn, topo.fix3, topo.nodal_loads, topo.wy, topo.wx, A=cfg.A, E=cfg.E, nu=cfg.nu,
                             I0=cfg.I0, alpha_moment=cfg.alpha_moment, alpha_shear=cfg.alpha_shear, k=cfg.k, num_epochs=5,
                             lr=cfg.lr, tolerance=cfg.tolerance, patience=cfg.patience, I_init=I0[b]) for b in range(B)]
    for n in (1, 2, 5):
        I, sol, ep = frames.optimize_frames(topo, B, cfg, I0=torch.as_tensor(I0, device="cuda"), max_epochs=n, poll_every=1)
        got = I.cpu().numpy()
        for b in range(B):
            np.testing.assert_allclose(got[b], ref[b]["I_history"][n - 1], rtol=2e-5 * n, atol=0)


def _build_grid_frame_through_the_shim(ops, cfg, bays, stories, I, lateral):
    nb1 = bays + 1
    coords = {i * nb1 + j + 1: (j * cfg.bay_width, i * cfg.story_height) for i in range(stories + 1) for j in range(nb1)}
    n_cols, n_beams = stories * nb1, stories * bays
    ops.wipe()
    ops.model('basic', '-ndm', 2, '-ndf', 3)
    ops.geomTransf('Linear', 1)
    for tag, (x, y) in coords.items():
        ops.node(tag, x, y)
    for tag, (x, y) in coords.items():
        if y == 0.0:
            ops.fix(tag, 1, 1, 1)
    e = 1
    for i in range(stories):
        for j in range(nb1):
            ops.element('elasticBeamColumn', e, i * nb1 + j + 1, (i + 1) * nb1 + j + 1, cfg.A, cfg.E, float(I[e - 1]), 1); e += 1
    for i in range(1, stories + 1):
        for j in range(bays):
            ops.element('elasticBeamColumn', e, i * nb1 + j + 1, i * nb1 + j + 2, cfg.A, cfg.E, float(I[e - 1]), 1); e += 1
    ops.timeSeries('Linear', 1); ops.pattern('Plain', 1, 1)
    for tag, (x, y) in coords.items():
        if x == 0.0 and y != 0.0:
            ops.load(tag, lateral, 0.0, 0.0)
    for ele in range(n_cols + 1, n_cols + n_beams + 1):
        ops.eleLoad('-ele', ele, '-type', '-beamUniform', cfg.vertical_load, cfg.vertical_load)
    ops.system('BandGeneral'); ops.numberer('RCM'); ops.constraints('Plain'); ops.integrator('LoadControl', 1.0)
    ops.algorithm('Newton'); ops.analysis('Static')
    return ops.analyze(1)


def test_ops_shim_deferred_batches_frames_by_topology():
    """`ops.deferred()` around the frame script's command sequence: frames of one topology (different inertias AND nodal
    loads) go out in one launch, a second topology in another; every queued domain gets its own results."""
    from openpystruct_amd import frames, ops
    cfg = frames.FrameConfig()
    rng = np.random.default_rng(3)
    specs = [(3, 2, 1.0e4), (3, 2, 2.5e4), (2, 2, 1.0e4), (3, 2, -0.7e4), (2, 2, 3.0e4)]
    Is = [np.full(s * (b + 1) + s * b, cfg.I0) * rng.uniform(0.5, 2.0, size=s * (b + 1) + s * b) for b, s, _ in specs]
    with ops.deferred() as batch:
        for (b, s, lat), I in zip(specs, Is):
            assert _build_grid_frame_through_the_shim(ops, cfg, b, s, I, lat) == 0
    assert batch.codes == [0] * len(specs) and len(batch.domains) == len(specs)
    for (b, s, lat), I, dom in zip(specs, Is, batch.domains):
        c2 = frames.FrameConfig(lateral_load=lat)
        topo = frames.grid_frame(b, s, c2, device="cpu")
        d, f, st, _, _ = _oracle(topo, I)
        np.testing.assert_allclose(dom.result["forces"], f, rtol=1e-7, atol=1e-6 * np.abs(f).max())
        np.testing.assert_allclose(dom.result["v"], d[:, 1], rtol=1e-7, atol=1e-12)


def test_tall_narrow_frame_falls_back_to_the_workgroup_per_frame_kernels():
    """3 bays x 420 stories: 5 040 equations at half bandwidth 14.  The wave-per-frame kernel keeps four right-hand sides in one CU's
    LDS and cannot hold them; the dispatch (and ops_frame_workspace_bytes, consistently) must fall back to the LDS-ring kernels."""
    from openpystruct_amd import frames
    topo = frames.grid_frame(3, 420)
    assert topo.n_eq == 3 * 4 * 420 and topo.kd == 14
    rng = np.random.default_rng(1)
    I = np.exp(rng.uniform(np.log(2e-3), np.log(5e-3), size=(2, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    torch.cuda.synchronize()
    assert int(sol.status.abs().sum()) == 0
    for b in range(2):
        d, f, st, neq, kd = _oracle(topo, I[b])
        assert st == 0
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-6      # a 420-story cantilever: cond ~ 1e10
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-6


def test_solves_on_one_topology_from_two_streams_do_not_share_a_workspace():
    from openpystruct_amd import frames
    topo = frames.grid_frame(10, 10)
    g = torch.Generator(device="cuda").manual_seed(3)
    Ia = torch.exp(torch.empty((4096, topo.Ne), dtype=torch.float64, device="cuda").uniform_(np.log(1e-4), np.log(5e-3), generator=g))
    Ib = torch.exp(torch.empty((4096, topo.Ne), dtype=torch.float64, device="cuda").uniform_(np.log(1e-4), np.log(5e-3), generator=g))
    ra = frames.frame_solve(topo, Ia)
    rb = frames.frame_solve(topo, Ib)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sa):
            qa = frames.frame_solve(topo, Ia)
        with torch.cuda.stream(sb):
            qb = frames.frame_solve(topo, Ib)
    torch.cuda.synchronize()
    # (not bit for bit: the assembly adds element contributions with LDS atomics, whose order varies from run to run)
    for q, r in ((qa, ra), (qb, rb)):
        assert relerr(q.disp.cpu().numpy().reshape(4096, -1), r.disp.cpu().numpy().reshape(4096, -1)) < 1e-10
        assert relerr(q.forces.cpu().numpy().reshape(4096, -1), r.forces.cpu().numpy().reshape(4096, -1)) < 1e-9
    assert len(topo._ws) >= 2


@pytest.mark.parametrize("bays,stories,B", [(15, 16, 12288), (10, 10, 16384)])
def test_config5_at_the_batch_the_bench_quotes(bays, stories, B):
    """bench.py launches 12 288 frames of 15 x 16 and 16 384 of 10 x 10, while every oracle comparison used <= 6 frames.
    Here, at those batches: 32 frames -- the first, the last, the frames either side of the point where the persistent waves (r06: one
    workspace slot per resident wave, each wave walking over its share of the batch) start their second turn, and seeded picks -- against the
    oracle's dpbsv solve; invariance under a permutation of the batch; two streams solving different halves' worth of inputs at full size
    without sharing factor columns."""
    from openpystruct_amd import _cabi, frames
    lib = _cabi.load()
    topo = frames.grid_frame(bays, stories)
    g = torch.Generator(device="cuda").manual_seed(20250307)
    I = torch.exp(torch.empty((B, topo.Ne), dtype=torch.float64, device="cuda").uniform_(np.log(1e-4), np.log(5e-3), generator=g))
    sol = frames.frame_solve(topo, I)
    torch.cuda.synchronize()
    assert int(sol.status.abs().sum()) == 0
    assert int(lib.ops_frame_workspace_bytes(B + 1, topo.n_eq, topo.kd)) > int(lib.ops_frame_workspace_bytes(B, topo.n_eq, topo.kd)) > 0
    pick = {0, 1, B - 2, B - 1}
    for waves_per_cu in (4, 8, 12, 16):                # resident waves = 256 CUs x (by LDS / registers) 4 .. 16: where a wave's second frame starts
        k = 256 * waves_per_cu
        if k + 1 < B:
            pick |= {k - 1, k, k + 1}
    rng = np.random.default_rng(B)
    while len(pick) < 32:
        pick.add(int(rng.integers(0, B)))
    pick = sorted(pick)
    Ih, dh, fh = I[pick].cpu().numpy(), sol.disp[pick].cpu().numpy(), sol.forces[pick].cpu().numpy()
    for j, b in enumerate(pick):
        d, f, st, neq, kd = _oracle(topo, Ih[j])
        assert st == 0
        assert relerr(dh[j].ravel(), d.ravel()) < 1e-7, b
        assert relerr(fh[j].ravel(), f.ravel()) < 1e-6, b
    # permutation of the batch: frame b's answer does not depend on where it sits (LDS atomics of the assembly: equal to rounding)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    solp = frames.frame_solve(topo, I[perm].contiguous())
    torch.cuda.synchronize()
    assert int(solp.status.abs().sum()) == 0
    dmax = sol.disp.abs().amax(dim=(1, 2))[perm]
    assert float(((solp.disp - sol.disp[perm]).abs().amax(dim=(1, 2)) / dmax).max()) < 1e-9
    fmax = sol.forces.abs().amax(dim=(1, 2))[perm]
    assert float(((solp.forces - sol.forces[perm]).abs().amax(dim=(1, 2)) / fmax).max()) < 1e-8
    del solp
    # two streams at full size on one topology: each has its own workspace (4.4 GB each at 15 x 16)
    I2 = I.flip(0).contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    for _ in range(2):
        with torch.cuda.stream(sa):
            qa = frames.frame_solve(topo, I)
        with torch.cuda.stream(sb):
            qb = frames.frame_solve(topo, I2)
    torch.cuda.synchronize()
    assert float(((qa.disp - sol.disp).abs().amax(dim=(1, 2)) / sol.disp.abs().amax(dim=(1, 2))).max()) < 1e-9
    assert float(((qb.disp - sol.disp.flip(0)).abs().amax(dim=(1, 2)) / sol.disp.flip(0).abs().amax(dim=(1, 2))).max()) < 1e-9
    topo._ws.clear()
    del qa, qb, sol
    torch.cuda.empty_cache()


def test_small_batches_take_a_workgroup_per_frame_and_large_ones_a_wave():
    """The dispatch itself (r05): <= 256 frames (and more of a small frame: the batch at which the two kernels meet, at most 4 000) need no
    factor workspace when the band fits LDS (workgroup-per-frame kernels), 8 193 do (wave-per-frame kernel); both agree with the oracle and
    with each other to rounding, frame by frame."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from openpystruct_amd import _cabi, frames
    _cabi.set_option("frame_latency_batch", -1)
    lib = _cabi.load()
    topo = frames.grid_frame(6, 5)
    assert int(lib.ops_frame_workspace_bytes(256, topo.n_eq, topo.kd)) == 0 and int(lib.ops_frame_workspace_bytes(8193, topo.n_eq, topo.kd)) > 0
    big10 = frames.grid_frame(10, 10)                     # the largest frame the script draws: the two meet at ~500 frames
    assert int(lib.ops_frame_workspace_bytes(256, big10.n_eq, big10.kd)) == 0 and int(lib.ops_frame_workspace_bytes(1024, big10.n_eq, big10.kd)) > 0
    rng = np.random.default_rng(256)
    I = np.exp(rng.uniform(np.log(1e-4), np.log(5e-3), size=(8193, topo.Ne)))
    Id = torch.as_tensor(I, device="cuda")
    big = frames.frame_solve(topo, Id)
    small = frames.frame_solve(topo, Id[:256].contiguous())
    assert int(big.status.abs().sum()) == 0 and int(small.status.abs().sum()) == 0
    a, b = big.disp[:256].cpu().numpy(), small.disp.cpu().numpy()
    assert np.abs(a - b).max() <= 1e-11 * np.abs(a).max()
    for k in (0, 100, 255):
        d, f, st, _, _ = _oracle(topo, I[k])
        assert st == 0
        for sol, kk in ((big, k), (small, k)):
            assert np.abs(sol.disp[kk].cpu().numpy() - d).max() <= 1e-7 * np.abs(d).max()
            assert np.abs(sol.forces[kk].cpu().numpy() - f).max() <= 1e-6 * np.abs(f).max()
    _cabi.set_option("frame_latency_batch", 0)
    assert int(lib.ops_frame_workspace_bytes(1, topo.n_eq, topo.kd)) > 0
    assert _cabi.get_option("frame_latency_batch") == 0 and _cabi.get_option("frame_pack") == 1
    with pytest.raises(ValueError):
        _cabi.set_option("no_such_option", 1)


# ---- r06: several frames per wavefront (csrc/frame_pack.hpp) ----
_PACK_SHAPES = [(1, 1, 5, 6), (10, 1, 5, 6), (1, 10, 8, 10), (2, 2, 8, 10), (3, 3, 11, 12), (21, 3, 11, 12), (4, 4, 14, 16), (9, 4, 14, 16),
                (5, 5, 17, 18), (10, 6, 20, 22), (7, 7, 23, 24), (8, 8, 26, 28), (10, 8, 26, 28), (9, 9, 29, 30), (8, 10, 29, 30)]


@pytest.mark.parametrize("bays,stories,kd,W", _PACK_SHAPES)
def test_packed_frames_vs_oracle_and_isolation(bays, stories, kd, W):
    """Half bandwidths up to 29 (98 of the 100 (bays, stories) draws of FR:17-18): 16 or 32 lanes per frame, 4 or 2 frames per wave -- every
    compiled (window width, lanes, group size) against the oracle, with a frame that is not positive definite and a frame with a NaN load in the
    SAME waves as healthy ones (nothing may cross between the lane groups of a wave), a batch that does not fill its last wave, and against one
    wave per frame (library option frame_pack = 0: same arithmetic, so the displacements agree to the order of the assembly's LDS additions)."""
    from openpystruct_amd import _cabi, frames
    topo = frames.grid_frame(bays, stories)
    assert topo.kd == kd
    lib = _cabi.load()
    B = 11
    sig = int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd))
    assert sig >> 24 == 2 and (sig >> 16) & 0xFF == W               # the packed kernel, this window width
    rng = np.random.default_rng(bays * 100 + stories)
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    I[6, min(3, topo.Ne - 1)] = -1.0
    loads = np.broadcast_to(topo.nodal_loads, (B,) + topo.nodal_loads.shape).copy()
    loads[9, -1, 0] = np.nan
    It, Lt = torch.as_tensor(I, device="cuda"), torch.as_tensor(loads, device="cuda")
    sol = frames.frame_solve(topo, It, Lt)
    torch.cuda.synchronize()
    st = sol.status.cpu().numpy()
    healthy = [b for b in range(B) if b not in (6, 9)]
    assert st[6] != 0 and st[healthy].sum() == 0 and torch.isnan(sol.disp[6]).all() and torch.isnan(sol.disp[9]).any()
    for b in healthy:
        d, f, s_, neq, _ = _oracle(topo, I[b])
        assert s_ == 0 and neq == topo.n_eq
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8, b
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7, b
    _cabi.set_option("frame_pack", 0)
    assert int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd)) >> 24 == 1
    ref = frames.frame_solve(topo, It, Lt)
    torch.cuda.synchronize()
    assert relerr(sol.disp[healthy].cpu().numpy().ravel(), ref.disp[healthy].cpu().numpy().ravel()) < 1e-11
    np.testing.assert_array_equal(ref.status.cpu().numpy() != 0, st != 0)


def test_packed_frames_keep_the_plan_between_calls(monkeypatch):
    """The assembly plan at the start of the workspace is built by the first call on a FrameTopology and kept (OPS_FRAME_REUSE_PLAN): later calls
    with other inertias, other loads and another batch size answer like the oracle; a batch size that changes the kernel family rebuilds; the plain
    C entry point (no flag) rebuilds every time and gives the same answers."""
    from openpystruct_amd import _cabi, frames
    topo = frames.grid_frame(5, 5)
    rng = np.random.default_rng(5)
    calls = []
    lib = _cabi.load()
    real = lib.ops_frame_solve_batched_f64_ex

    class Spy:
        def __call__(self, *a):
            calls.append(a[-1])
            return real(*a)
    monkeypatch.setattr(lib, "ops_frame_solve_batched_f64_ex", Spy())
    for B in (9, 9, 4, 9):
        I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
        loads = np.broadcast_to(topo.nodal_loads, (B,) + topo.nodal_loads.shape) * rng.uniform(0.5, 2.0, size=(B, 1, 1))
        sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"), torch.as_tensor(loads.copy(), device="cuda"))
        torch.cuda.synchronize()
        assert int(sol.status.abs().sum()) == 0
        for b in (0, B - 1):
            d, f, s_, _, _ = bo.solve_model_3dof(topo.coords, topo.conn, topo.A, topo.E, I[b], topo.fix3, loads[b], wy=topo.wy, wx=topo.wx)
            assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
            assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7
    assert calls == [0, _cabi.FRAME_REUSE_PLAN, _cabi.FRAME_REUSE_PLAN, _cabi.FRAME_REUSE_PLAN]
    # the library's default dispatch sends small batches to the workgroup-per-frame kernels (no plan): the signature changes, the flag goes
    _cabi.set_option("frame_latency_batch", -1)
    assert int(lib.ops_frame_plan_signature(9, topo.n_eq, topo.kd)) == 0 and int(lib.ops_frame_plan_signature(8192, topo.n_eq, topo.kd)) != 0


def test_packed_frames_persistent_waves_walk_the_batch():
    """More frames than the chip holds at once: every wave solves several sets of frames with ONE workspace slot (the launch is capped at the
    resident workgroups).  Sampled frames against the oracle, and the whole batch against itself in another order (a frame's answer may not depend
    on which wave, lane group or turn of the loop solves it)."""
    from openpystruct_amd import frames
    topo = frames.grid_frame(2, 2)
    B = 70001                                           # 16 frames per workgroup: > 4 000 workgroups, a last wave with one frame
    rng = np.random.default_rng(22)
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    It = torch.as_tensor(I, device="cuda")
    sol = frames.frame_solve(topo, It)
    assert int(sol.status.abs().sum()) == 0
    for b in (0, 1, 15, 16, 4095, 16384, 40000, 65535, 65536, B - 2, B - 1):
        d, f, s_, _, _ = _oracle(topo, I[b])
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8, b
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7, b
    perm = torch.randperm(B, device="cuda")
    sol2 = frames.frame_solve(topo, It[perm].contiguous())
    scale = float(sol.disp.abs().max())
    assert float((sol2.disp - sol.disp[perm]).abs().max()) < 1e-11 * scale


@pytest.mark.parametrize("bays,kd", [(41, 128), (99, 302), (339, 1022)])
def test_column_by_column_fallback_up_to_the_half_bandwidth_it_claims(bays, kd):
    """`frame_wide_kernel` (csrc/frame_solve.hip; the reference's BandGeneral has no limit, FR:134) says it serves half bandwidths up to 1 024:
    two-story frames numbered story by story at 128, 302 and 1 022 (the largest a grid can have below the limit: its backward sweep holds a column
    in 16 registers per lane of one wave) against the oracle, with a frame that is not positive definite between two healthy ones."""
    from openpystruct_amd import frames
    topo = frames.grid_frame(bays, 2, numbering="node")
    assert topo.kd == kd and topo.n_eq == 6 * (bays + 1)
    rng = np.random.default_rng(kd)
    I = np.exp(rng.uniform(np.log(1e-4), np.log(5e-3), size=(3, topo.Ne)))
    I[1, 5] = -1.0
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    st = sol.status.cpu().numpy()
    assert st[0] == 0 and st[2] == 0 and st[1] != 0 and torch.isnan(sol.disp[1]).all()
    for k in (0, 2):
        d, f, s_, neq, okd = _oracle(topo, I[k])
        assert s_ == 0 and okd == kd
        assert np.abs(sol.disp[k].cpu().numpy() - d).max() <= 1e-7 * np.abs(d).max(), (kd, k)
        assert np.abs(sol.forces[k].cpu().numpy() - f).max() <= 1e-6 * np.abs(f).max(), (kd, k)


def test_half_bandwidth_beyond_the_fallback_is_refused_cleanly():
    """The first half bandwidth past the limit (340 bays story by story: 1 025) is OPS_AMD_ERR_UNSUPPORTED -- NotImplementedError on the host
    side --, not a fault, and the next call on the same stream still answers."""
    from openpystruct_amd import _cabi, frames
    topo = frames.grid_frame(340, 2, numbering="node")
    assert topo.kd == 1025
    I = torch.full((2, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
    with pytest.raises(NotImplementedError):
        frames.frame_solve(topo, I)
    lib = _cabi.load()
    ws = torch.empty(int(lib.ops_frame_workspace_bytes(2, topo.n_eq, topo.kd)), dtype=torch.uint8, device="cuda")
    outs = [torch.empty(s, dtype=torch.float64, device="cuda") for s in ((2, topo.Nn, 3), (2, topo.Ne, 6), (2, topo.Ne), (2, topo.Ne))]
    st = torch.zeros(2, dtype=torch.int32, device="cuda")
    rc = lib.ops_frame_solve_batched_f64(2, topo.Nn, topo.Ne, topo.n_eq, topo.kd, topo.d_geo.data_ptr(), topo.d_EA.data_ptr(), topo.d_E.data_ptr(),
                                         topo.d_w.data_ptr(), topo.d_elem_eq.data_ptr(), topo.d_node_eq.data_ptr(), I.data_ptr(), topo.d_loads.data_ptr(), 0,
                                         *(o.data_ptr() for o in outs), st.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == _cabi.ERR_UNSUPPORTED
    ok = frames.frame_solve(frames.grid_frame(2, 2), torch.full((2, 10), 5e-4, dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()
    assert int(ok.status.abs().sum()) == 0 and bool(torch.isfinite(ok.disp).all())


@pytest.mark.parametrize("bays,stories,family", [(60, 3, 2), (300, 3, 1)])
def test_long_narrow_frames_take_the_fallbacks_of_the_packed_kernel(bays, stories, family):
    """Half bandwidth 11 with hundreds to thousands of equations: (60, 3) still fits the packed kernel's LDS but not with its inertias staged there
    (363 elements x 16 frames per workgroup: gathered from HBM instead, frame_pack.hpp `stage_I`); (300, 3) -- 2 709 equations -- does not fit
    sixteen frames' solution vectors at all and takes the wave-per-frame kernel at its narrowest window (36).  Both against the oracle."""
    from openpystruct_amd import _cabi, frames
    topo = frames.grid_frame(bays, stories)
    assert topo.kd == 11
    B = 7
    assert int(_cabi.load().ops_frame_plan_signature(B, topo.n_eq, topo.kd)) >> 24 == family
    rng = np.random.default_rng(bays)
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    I[3, 2] = -1.0
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    st = sol.status.cpu().numpy()
    assert st[3] != 0 and st[[0, 1, 2, 4, 5, 6]].sum() == 0
    for b in (0, 2, 4, 6):
        d, f, s_, neq, _ = _oracle(topo, I[b])
        assert s_ == 0 and neq == topo.n_eq
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-7, b
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-6, b


def test_every_draw_of_the_reference_range_at_a_batch_the_tuned_kernels_serve():
    """FR:17-18, 50-52: bays, stories ~ U{1..10} -- all 100 frames of that range, each as a batch of 300 frames (beyond the library's small-batch
    threshold wherever that threshold is 256, forced for the others by the option this file sets: every shape goes to the packed or the
    wave-per-frame kernel), two frames of each batch against the oracle, plus the batch's own consistency: frames 0 and 299 carry the same inertias
    and must answer bit for bit alike whichever wave and lane group they land in.  The worst deviations per kernel family go to gpurun_out/."""
    import json
    import os
    from openpystruct_amd import _cabi, frames
    lib = _cabi.load()
    worst = {}
    fams = {1: "wave", 2: "packed"}
    B = 300
    for bays in range(1, 11):
        for stories in range(1, 11):
            topo = frames.grid_frame(bays, stories)
            sig = int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd))
            fam = fams[sig >> 24]
            assert fam == ("packed" if topo.kd <= 29 else "wave"), (bays, stories, topo.kd)
            rng = np.random.default_rng(1000 * bays + stories)
            I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
            I[B - 1] = I[0]
            sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
            torch.cuda.synchronize()
            assert int(sol.status.abs().sum()) == 0, (bays, stories)
            assert torch.equal(sol.disp[0], sol.disp[B - 1]) and torch.equal(sol.forces[0], sol.forces[B - 1]), (bays, stories)
            for b in (0, 157):
                d, f, st, neq, _ = _oracle(topo, I[b])
                assert st == 0 and neq == topo.n_eq
                ed = relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel())
                ef = relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel())
                assert ed < 1e-8 and ef < 1e-7, (bays, stories, b, ed, ef)
                w = worst.setdefault(fam, {"shapes": 0, "disp": 0.0, "forces": 0.0, "at": None})
                if ed > w["disp"]:
                    w["disp"], w["at"] = ed, [bays, stories]
                w["forces"] = max(w["forces"], ef)
            worst[fam]["shapes"] += 1
    assert worst["packed"]["shapes"] == 98 and worst["wave"]["shapes"] == 2
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/frames_full_range.json", "w") as fh:
        json.dump(worst, fh, indent=1)


@pytest.mark.parametrize("bays,stories,B", [(2, 2, 3), (2, 2, 5), (5, 5, 1), (5, 5, 7), (9, 9, 3)])
def test_packed_frames_stay_inside_the_workspace_the_library_asks_for(bays, stories, B):
    """A batch that does not fill its last wave: the lane groups past the end solve the last frame once more into their OWN factor slot (r06: every
    window bound is then wave-uniform), so ops_frame_workspace_bytes counts whole waves.  The call runs on a buffer of exactly that size followed
    by a guard band; the guard must come back untouched (the first version of that change wrote one frame slot past a B-frame workspace)."""
    from openpystruct_amd import _cabi, frames
    topo = frames.grid_frame(bays, stories)
    lib = _cabi.load()
    assert int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd)) >> 24 == 2
    need = int(lib.ops_frame_workspace_bytes(B, topo.n_eq, topo.kd))
    guard = 1 << 20
    buf = torch.full((need + guard,), 0xA5, dtype=torch.uint8, device="cuda")
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    topo.__dict__["_ws"] = {key: [buf, 0]}            # frames.frame_solve takes a cached buffer that is large enough and passes `need` as its size
    rng = np.random.default_rng(7)
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    torch.cuda.synchronize()
    assert topo.__dict__["_ws"][key][0] is buf
    assert bool((buf[need:] == 0xA5).all()), "the solve wrote past the workspace it asked for"
    assert int(sol.status.abs().sum()) == 0
    d, f, st, _, _ = _oracle(topo, I[B - 1])
    assert relerr(sol.disp[B - 1].cpu().numpy().ravel(), d.ravel()) < 1e-8


@pytest.mark.parametrize("nn,hub,G,epg", [(9, 4, 4, 64), (12, 6, 8, 128), (14, 8, 8, 128), (18, 9, 4, 64), (18, 10, 2, 32)])
def test_packed_kernel_walks_the_extra_plan_blocks_of_a_hub_node(nn, hub, G, epg):
    """The packed kernel's plan blocks hold 16 G entries per row group (frame_pack.hpp fp_epg: what a four-element node needs); a node tied to
    EVERY other node of a short chain overflows them at every group size (sixteen lanes: half bandwidth 11; G = 8 / 4 / 2 <-> 17 and 23 / 26 / 29), so the solve walks the
    group's extra blocks -- the packed counterpart of the eighteen-element hub test above (which lands on the wave kernel)."""
    from openpystruct_amd import _cabi, frames
    coords = np.array([(1.2 * i, 0.5 * np.cos(1.3 * i) + (0.8 if i == hub else 0.0)) for i in range(nn)])
    conn = np.array([(i, i + 1) for i in range(nn - 1) if hub not in (i, i + 1)] + [(hub, o) for o in range(nn) if o != hub])
    fix3 = np.zeros((nn, 3), dtype=bool)
    fix3[0] = fix3[nn - 1] = True
    loads = np.zeros((nn, 3)); loads[hub] = (2e4, -3e4, 1e3); loads[2] = (0.0, -1e4, 0.0)
    w = np.zeros(len(conn)); w[:3] = -6e3
    topo = frames.FrameTopology(coords, conn, fix3, 0.02, 200e9, w, 0.5 * w, loads, "cuda", numbering="node")
    sig = int(_cabi.load().ops_frame_plan_signature(300, topo.n_eq, topo.kd))
    assert sig >> 24 == 2 and (sig >> 8) & 0xFF == G, (topo.kd, hex(sig))
    eq = topo.d_elem_eq.cpu().numpy()
    per_group = np.zeros(topo.n_eq // G + 1, dtype=int)
    for e in range(topo.Ne):
        for r in range(6):
            if eq[e, r] >= 0:
                per_group[eq[e, r] // G] += int(((eq[e] >= 0) & (eq[e] <= eq[e, r])).sum())
    assert per_group.max() > epg, per_group.max()
    rng = np.random.default_rng(nn)
    B = 7
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
    assert int(sol.status.abs().sum()) == 0
    for b in range(B):
        d, f, st, neq, kd = _oracle(topo, I[b])
        assert st == 0 and neq == topo.n_eq and _kd_ok(topo, kd)
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7


@pytest.mark.parametrize("bays,stories,W", [(1, 1, 20), (2, 2, 20), (5, 5, 20), (7, 7, 36), (10, 10, 36), (12, 12, 52), (15, 16, 52), (20, 17, 56)])
def test_four_waves_per_frame_vs_oracle(bays, stories, W):
    """csrc/frame_coop.hpp (r06, late): a workgroup of four waves per frame, the register window split by columns -- the small-batch kernel where
    the r01 kernels' band is not LDS-resident or fits a CU only once.  Library option frame_coop = 2 sends EVERY small batch to it: each compiled
    window width (20, 36, 52, 56) against the oracle, one frame that is not positive definite and one with a NaN load in the batch (status and
    NaN stay in their frames), and against the r01 kernels on the same inputs."""
    from openpystruct_amd import _cabi, frames
    _cabi.set_option("frame_latency_batch", -1)
    _cabi.set_option("frame_coop", 2)
    topo = frames.grid_frame(bays, stories)
    lib = _cabi.load()
    B = 6
    sig = int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd))
    assert sig >> 24 == 3 and (sig >> 16) & 0xFF == W, hex(sig)
    rng = np.random.default_rng(31 * bays + stories)
    I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
    I[2, min(3, topo.Ne - 1)] = -1.0
    loads = np.broadcast_to(topo.nodal_loads, (B,) + topo.nodal_loads.shape).copy()
    loads[4, -1, 0] = np.nan
    It, Lt = torch.as_tensor(I, device="cuda"), torch.as_tensor(loads, device="cuda")
    sol = frames.frame_solve(topo, It, Lt)
    torch.cuda.synchronize()
    st = sol.status.cpu().numpy()
    healthy = [0, 1, 3, 5]
    assert st[2] != 0 and st[healthy].sum() == 0 and torch.isnan(sol.disp[2]).all() and torch.isnan(sol.disp[4]).any()
    for b in healthy:
        d, f, s_, neq, _ = _oracle(topo, I[b])
        assert s_ == 0 and neq == topo.n_eq
        assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8, b
        assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7, b
    _cabi.set_option("frame_coop", 0)
    assert int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd)) == 0
    ref = frames.frame_solve(topo, It, Lt)
    torch.cuda.synchronize()
    assert relerr(sol.disp[healthy].cpu().numpy().ravel(), ref.disp[healthy].cpu().numpy().ravel()) < 1e-10
    np.testing.assert_array_equal(ref.status.cpu().numpy() != 0, st != 0)


def test_four_waves_per_frame_on_general_topologies_and_a_hub_node():
    """Pinned bases, braces (equation count no multiple of three or eight) and a node with eighteen elements (extra plan blocks) through
    csrc/frame_coop.hpp; a second call on the same topology reuses the plan."""
    from openpystruct_amd import _cabi, frames
    _cabi.set_option("frame_latency_batch", -1)
    _cabi.set_option("frame_coop", 2)
    topos = [_custom_frame(2, 2, True, True), _custom_frame(3, 4, True, True), _custom_frame(6, 5, True, False), _custom_frame(9, 9, True, True)]
    nn, hub = 19, 9
    ang = np.linspace(0.0, 2 * np.pi, nn - 1, endpoint=False)
    coords = np.zeros((nn, 2))
    outer = [i for i in range(nn) if i != hub]
    coords[outer, 0], coords[outer, 1] = 5.0 * np.cos(ang), 5.0 * np.sin(ang)
    conn = np.array([(hub, o) for o in outer] + [(outer[i], outer[i + 1]) for i in range(len(outer) - 1)])
    fix3 = np.zeros((nn, 3), dtype=bool)
    fix3[0] = fix3[nn - 1] = True
    loads = np.zeros((nn, 3)); loads[hub] = (3e4, -5e4, 2e3)
    w = np.zeros(len(conn)); w[:4] = -8e3
    topos.append(frames.FrameTopology(coords, conn, fix3, 0.02, 200e9, w, 0.5 * w, loads, "cuda", numbering="node"))
    lib = _cabi.load()
    for topo in topos:
        if topo.kd > 55:
            continue
        B = 3
        assert int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd)) >> 24 == 3
        rng = np.random.default_rng(topo.n_eq)
        for rep in range(2):                                   # second call: the plan kept at the start of the workspace
            I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
            sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"))
            assert int(sol.status.abs().sum()) == 0
            for b in range(B):
                d, f, st, neq, kd = _oracle(topo, I[b])
                assert st == 0 and neq == topo.n_eq
                assert relerr(sol.disp[b].cpu().numpy().ravel(), d.ravel()) < 1e-8
                assert relerr(sol.forces[b].cpu().numpy().ravel(), f.ravel()) < 1e-7


def test_small_batch_dispatch_picks_the_family_the_sweep_measured():
    """profiles/r06_frame_coop_sweep*.txt: up to a frame per CU the r01 kernels where their band is LDS-resident (10 x 10), four waves per frame
    where it is not (15 x 16, up to 512 frames) and between one and three frames per CU where the r01 kernel fits a CU only once (10 x 10,
    257 .. 768 frames); the tuned kernels beyond.  Host-side functions only."""
    from openpystruct_amd import _cabi, frames
    _cabi.set_option("frame_latency_batch", -1)
    lib = _cabi.load()
    fam = lambda t, B: int(lib.ops_frame_plan_signature(B, t.n_eq, t.kd)) >> 24       # 0: r01 workgroup kernels, 1: wave, 2: packed, 3: four waves per frame
    t10, t15, t5, t9 = (frames.grid_frame(*s, device="cpu") for s in ((10, 10), (15, 16), (5, 5), (9, 9)))
    assert [fam(t10, B) for B in (1, 256, 257, 512, 768, 769, 4096)] == [0, 0, 3, 3, 3, 1, 1]
    assert [fam(t15, B) for B in (1, 256, 512, 513, 12288)] == [3, 3, 3, 1, 1]
    assert [fam(t5, B) for B in (1, 256, 512, 1024, 4096)] == [0, 0, 0, 0, 2]
    assert fam(t9, 256) == 0 and fam(t9, 4096) == 2
    _cabi.set_option("frame_coop", 0)
    assert [fam(t10, B) for B in (257, 512)] == [0, 1] and fam(t15, 64) == 0
    _cabi.set_option("frame_coop", 1)
