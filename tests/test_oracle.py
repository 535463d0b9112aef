"""Pins the CPU oracle: closed-form Euler-Bernoulli known answers, two independent
formulations (2-DOF dense vs OpenSees-like 3-DOF banded dpbsv), the plain-C restatement,
equilibrium identities, and the committed golden fixtures.  (PARITY UNPINNED w.r.t. a live
OpenSeesPy: see oracle/beam_oracle.py header.)"""
import os

import numpy as np
import pytest

from oracle import beam_oracle as bo
from oracle import c_oracle as co
from tests.helpers import load_golden, relerr

E = bo.E_REF


def _ss(N=101, L=200.0, EI=1e11):
    x = np.linspace(0, L, N)
    fix = np.zeros(N, dtype=np.uint8)
    fix[0] = fix[-1] = 1
    return x, fix, np.full(N - 1, EI / E)


def test_kat_simply_supported_udl():
    # SURVEY Appendix A.4 sign check: w=-1000, L=200, EI=1e11
    x, fix, I = _ss()
    w, L, EI = -1000.0, 200.0, 1e11
    v, th, V, M, st = bo.solve_beam_dense(x, E, I, fix, np.zeros(101), w)
    assert st == 0
    assert v[50] == pytest.approx(5 * w * L**4 / (384 * EI), rel=2e-8)
    assert th[0] == pytest.approx(w * L**3 / (24 * EI), rel=2e-8)
    assert V[0] == pytest.approx(-w * L / 2, rel=2e-8)          # +100 000: upward reaction
    assert M[50] == pytest.approx(w * L**2 / 8, rel=2e-8)       # -5.0e6 at end I of element 51
    assert abs(M[0]) < 1e-3


def test_kat_simply_supported_midspan_point_load():
    x, fix, I = _ss()
    P, L, EI = -1.0e5, 200.0, 1e11
    Fy = np.zeros(101); Fy[50] = P
    v, th, V, M, st = bo.solve_beam_dense(x, E, I, fix, Fy, 0.0)
    assert v[50] == pytest.approx(P * L**3 / (48 * EI), rel=2e-8)
    assert th[0] == pytest.approx(P * L**2 / (16 * EI), rel=2e-8)
    assert V[0] == pytest.approx(-P / 2, rel=2e-8)
    assert M[50] == pytest.approx(P * L / 4, rel=2e-8)


def test_kat_cantilever_tip_load():
    # clamped at node 1 (u_y and theta_z fixed), tip load: PL^3/3EI, PL^2/2EI
    N, L, EI, P = 41, 10.0, 2.0e7, -3.0e3
    x = np.linspace(0, L, N)
    fix = np.zeros(N, dtype=np.uint8); fix[0] = 3
    Fy = np.zeros(N); Fy[-1] = P
    v, th, V, M, st = bo.solve_beam_dense(x, E, np.full(N - 1, EI / E), fix, Fy, 0.0)
    assert v[-1] == pytest.approx(P * L**3 / (3 * EI), rel=2e-8)
    assert th[-1] == pytest.approx(P * L**2 / (2 * EI), rel=2e-8)
    assert V[0] == pytest.approx(-P, rel=2e-8)
    assert M[0] == pytest.approx(-P * L, rel=2e-8)   # resisting moment at the clamp (end I)


def test_kat_propped_cantilever_udl():
    # clamped left, roller right, UDL: R_roller = 3wL/8, max |M| at clamp = wL^2/8
    N, L, EI, w = 65, 16.0, 5.0e8, -2.0e3
    x = np.linspace(0, L, N)
    fix = np.zeros(N, dtype=np.uint8); fix[0] = 3; fix[-1] = 1
    v, th, V, M, st = bo.solve_beam_dense(x, E, np.full(N - 1, EI / E), fix, np.zeros(N), w)
    assert V[0] == pytest.approx(-5 * w * L / 8, rel=2e-8)
    assert M[0] == pytest.approx(-w * L**2 / 8, rel=2e-8)          # hogging at the clamp: positive at end I
    assert th[-1] == pytest.approx(-w * L**3 / (48 * EI), rel=2e-8)


def test_kat_two_span_continuous_udl():
    # two equal spans l, UDL: middle reaction 10wl/8, end reactions 3wl/8, M over middle support = wl^2/8
    N, l, EI, w = 81, 20.0, 3.0e9, -1.5e3
    x = np.linspace(0, 2 * l, N)
    fix = np.zeros(N, dtype=np.uint8); fix[0] = fix[40] = fix[-1] = 1
    v, th, V, M, st = bo.solve_beam_dense(x, E, np.full(N - 1, EI / E), fix, np.zeros(N), w)
    assert V[0] == pytest.approx(-3 * w * l / 8, rel=2e-8)
    assert M[40] == pytest.approx(-w * l**2 / 8, rel=2e-8)   # hogging: positive at end I
    assert th[40] == pytest.approx(0.0, abs=1e-12)


def test_kat_overhang_tip():
    # reference bridge probe of SURVEY Appendix E
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    Fy = np.zeros(101); Fy[49] = -355857.0; Fy[19] = -100000.0
    v, th, V, M, st = bo.solve_beam_dense(x, E, np.full(100, 0.5), fix, Fy, -1000.0)
    assert v.min() == pytest.approx(-1.634e-2, rel=1e-3)
    assert v[100] == pytest.approx(9.51e-5, rel=1e-3)
    assert th[100] == pytest.approx(4.76e-5, rel=2e-3)
    np.testing.assert_allclose(V[:3], [36417.18, 34417.18, 32417.18], rtol=1e-6)
    # overhang element 100: statically determinate, V = -w*L_e... free tip
    assert M[99] == pytest.approx(-(-1000.0) * 2.0**2 / 2, rel=1e-6)   # statically determinate overhang: M at end I = -w*a^2/2


def test_equilibrium_and_reactions():
    rng = np.random.default_rng(5)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, 8, inertia="trajectory")
    for b in range(8):
        K, f = bo.assemble_beam(x, E, I[b], Fy[b], bo.UDL_REF)
        v, th, V, M, st = bo.solve_beam_dense(x, E, I[b], fix, Fy[b], bo.UDL_REF)
        u = np.empty(202); u[0::2] = v; u[1::2] = th
        r = K @ u - f
        free = np.ones(202, dtype=bool); free[0::2] = fix == 0
        assert np.abs(r[free]).max() < 1e-6 * np.abs(f).max()
        # reactions balance the applied load (sum Fy + UDL*L)
        assert r[~free].sum() == pytest.approx(-(Fy[b].sum() + bo.UDL_REF * 200.0), rel=2e-8)


@pytest.mark.parametrize("inertia,tol", [("uniform", 1e-10), ("trajectory", 2e-9)])
def test_dense_vs_opensees_like_3dof(inertia, tol):
    """2-DOF dense solve == 3-DOF/node banded dpbsv model with axial DOFs and Wx (296 eq, kd 5)."""
    rng = np.random.default_rng(11)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, 6, inertia=inertia)
    for b in range(6):
        nodes = np.nonzero(Fy[b])[0] + 1
        d, f, st, neq, kd = bo.solve_reference_beam_3dof(x, bo.A_REF, E, I[b], bo.ROLLERS_REF, nodes, Fy[b, nodes - 1], bo.UDL_REF)
        assert (st, neq, kd) == (0, 296, 5)
        v, th, V, M, st2 = bo.solve_beam_dense(x, E, I[b], fix, Fy[b], bo.UDL_REF)
        assert relerr(d[:, 1], v) < tol and relerr(d[:, 2], th) < tol
        assert relerr(f[:, 1], V) < tol * 10 and relerr(f[:, 2], M) < tol * 10
        # axial UDL quirk (SingleCore.py:117): Fx at node I of element 1 = -N + p0 = +w*L_total... only forces[0] changes
        assert f[0, 0] == pytest.approx(-bo.UDL_REF * 200.0, rel=2e-8)


@pytest.mark.parametrize("name,tol", [("bridge_uniform", 1e-10), ("bridge_trajectory", 5e-9), ("random_bridge", 5e-8)])
def test_c_oracle_vs_golden(golden_dir, name, tol):
    g = load_golden(os.path.join(golden_dir, name + ".npz"))
    v, th, V, M, st = co.solve_beam_batched(g["x"], g["E"], g["I"], g["fix"], g["Fy"], g["wy"], n_threads=2)
    assert (st == g["status"]).all() and (st == 0).all()
    assert relerr(v, g["v"]) < tol and relerr(th, g["theta"]) < tol
    assert relerr(V, g["V"]) < tol * 10 and relerr(M, g["M"]) < tol * 10


def test_c_oracle_adversarial_cond_aware(golden_dir):
    # cond(K) up to ~3e8: tolerance scaled accordingly, reported separately
    g = load_golden(os.path.join(golden_dir, "bridge_adversarial.npz"))
    v, th, V, M, st = co.solve_beam_batched(g["x"], g["E"], g["I"], g["fix"], g["Fy"], g["wy"])
    assert relerr(v, g["v"]) < 1e-4 and relerr(th, g["theta"]) < 1e-4


def test_singular_reports_status():
    # non-positive pivot -> not SPD -> status != 0 (analyze() != 0, MultiCore.py:184).  (A pure
    # mechanism's last pivot is only ~0 in floating point, so dpbsv cannot be relied on to flag it.)
    x = np.linspace(0, 10, 11)
    fix = np.zeros(11, dtype=np.uint8); fix[0] = fix[-1] = 1
    I = np.full((3, 10), 0.1); I[0, :] = 0.0; I[1, 4] = -0.1
    Fy = np.zeros((3, 11)); Fy[:, 5] = -1.0
    st_np = bo.solve_beam_batched(x, E, I, fix, Fy, 0.0)[4]
    st_c = co.solve_beam_batched(x, E, I, fix, Fy, 0.0)[4]
    assert st_np[0] != 0 and st_np[1] != 0 and st_np[2] == 0
    assert st_c[0] != 0 and st_c[1] != 0 and st_c[2] == 0


def test_kat_reference_bridge_by_the_three_moment_equation():
    """The reference's own bridge (pin at x = 0, rollers at nodes 10, 30, 70, 85, 100 of a 200 m / 100-element beam, 2 m
    overhang; SingleCore.py:58-62) under its UDL alone, uniform EI: support moments from Clapeyron's three-moment equation
    -- an analysis that shares nothing with the FE formulation -- against the oracle's element end moments and reactions."""
    q = 1000.0                                            # |uniform_udl|, downward
    xs = np.array([0.0, 18.0, 58.0, 138.0, 168.0, 198.0])  # supports: 2 m * (node - 1)
    Ls = np.diff(xs)                                      # spans 18, 40, 80, 30, 30
    a = 200.0 - xs[-1]                                    # overhang
    M0, M5 = 0.0, -q * a * a / 2.0                        # sagging positive: free rotation at the pin, cantilever root moment
    # M_{i-1} L_i + 2 M_i (L_i + L_{i+1}) + M_{i+1} L_{i+1} = -q (L_i^3 + L_{i+1}^3) / 4,  i = 1..4
    A = np.zeros((4, 4)); rhs = np.zeros(4)
    for k, i in enumerate(range(1, 5)):
        Ll, Lr = Ls[i - 1], Ls[i]
        A[k, k] = 2.0 * (Ll + Lr)
        if k > 0: A[k, k - 1] = Ll
        if k < 3: A[k, k + 1] = Lr
        rhs[k] = -q * (Ll ** 3 + Lr ** 3) / 4.0
    rhs[3] -= Ls[4] * M5
    rhs[0] -= Ls[0] * M0
    Ms = np.concatenate([[M0], np.linalg.solve(A, rhs), [M5]])       # bending moments over the six supports
    # reactions from span equilibrium: shear just right / left of each support
    Vr = [q * Ls[i] / 2.0 + (Ms[i + 1] - Ms[i]) / Ls[i] for i in range(5)]          # left end of span i (upward on the beam)
    Vl = [q * Ls[i] / 2.0 - (Ms[i + 1] - Ms[i]) / Ls[i] for i in range(5)]          # right end of span i
    R = np.array([Vr[0]] + [Vl[i - 1] + Vr[i] for i in range(1, 5)] + [Vl[4] + q * a])
    assert R.sum() == pytest.approx(q * 200.0, rel=1e-12)

    x = np.linspace(0.0, 200.0, 101)
    fix = bo.reference_fix_mask()
    EI = bo.E_REF * bo.I0_REF
    v, th, V, M, st = bo.solve_beam_dense(x, bo.E_REF, np.full(100, bo.I0_REF), fix, np.zeros(101), -q)
    assert st == 0
    nodes = [1, 10, 30, 70, 85, 100]                      # 1-based support nodes
    # eleResponse 'forces'[2] at end I of element e (= node e) is MINUS the sagging-positive bending moment there
    for n, Mb in zip(nodes[1:], Ms[1:]):
        assert -M[n - 1] == pytest.approx(Mb, rel=2e-8), (n, -M[n - 1], Mb)
    # reactions: jump of the element end shears across a support node (forces[1] of element n minus -forces of element n-1)
    d3, f3, st3, _, _ = bo.solve_reference_beam_3dof(x, bo.A_REF, bo.E_REF, np.full(100, bo.I0_REF), bo.ROLLERS_REF, [], [], -q)
    Fy1, Fy2 = f3[:, 1], f3[:, 4]
    for n, Rn in zip(nodes, R):
        react = Fy1[n - 1] + (Fy2[n - 2] if n >= 2 else 0.0)    # sum of the element end forces meeting at the node = support reaction
        assert react == pytest.approx(Rn, rel=2e-8), (n, react, Rn)
    assert abs(EI) > 0


def test_kat_reference_bridge_with_point_loads_by_the_three_moment_equation():
    """Same bridge with the UDL plus two of the generator's point loads (SingleCore.py:157-160) at nodes 20 and 50."""
    q = 1000.0
    xs = np.array([0.0, 18.0, 58.0, 138.0, 168.0, 198.0])
    Ls = np.diff(xs)
    a_ov = 2.0
    loads = [(20, 3.0e5), (50, 1.2e5)]                    # (node, downward magnitude)
    M0, M5 = 0.0, -q * a_ov * a_ov / 2.0

    def span_terms(i):
        """6 A abar / L (moment-area term seen from the span's LEFT support) and 6 A bbar / L (from its RIGHT one)."""
        L = Ls[i]
        tl = tr = q * L ** 3 / 4.0
        for n, P in loads:
            xp = 2.0 * (n - 1)
            if xs[i] < xp < xs[i + 1]:
                a, b = xp - xs[i], xs[i + 1] - xp
                tl += P * a * (L * L - a * a) / L
                tr += P * b * (L * L - b * b) / L
        return tl, tr

    A = np.zeros((4, 4)); rhs = np.zeros(4)
    for k, i in enumerate(range(1, 5)):
        Ll, Lr = Ls[i - 1], Ls[i]
        A[k, k] = 2.0 * (Ll + Lr)
        if k > 0: A[k, k - 1] = Ll
        if k < 3: A[k, k + 1] = Lr
        rhs[k] = -(span_terms(i - 1)[0] + span_terms(i)[1])
    rhs[3] -= Ls[4] * M5
    Ms = np.concatenate([[M0], np.linalg.solve(A, rhs), [M5]])
    x = np.linspace(0.0, 200.0, 101)
    Fy = np.zeros(101)
    for n, P in loads:
        Fy[n - 1] -= P
    v, th, V, M, st = bo.solve_beam_dense(x, bo.E_REF, np.full(100, bo.I0_REF), bo.reference_fix_mask(), Fy, -q)
    assert st == 0
    for n, Mb in zip([10, 30, 70, 85, 100], Ms[1:]):
        assert -M[n - 1] == pytest.approx(Mb, rel=2e-8), (n, -M[n - 1], Mb)


def test_kat_frame_oracle_l_shaped_cantilever():
    """The 3-DOF frame oracle on a statically determinate L: column (0,0)-(0,h) clamped at its base, arm (0,h)-(a,h), vertical
    tip load P.  Closed form: column under axial P and constant moment P a; arm as a cantilever on the rotated column top."""
    h, a, P = 4.0, 3.0, -2.0e4
    E, A, I = 2.0e11, 8.0e-3, 3.0e-5
    nc, na = 8, 6
    coords = [(0.0, h * i / nc) for i in range(nc + 1)] + [(a * j / na, h) for j in range(1, na + 1)]
    conn = [(i, i + 1) for i in range(nc + na)]
    fix3 = np.zeros((len(coords), 3), dtype=np.int64); fix3[0] = 1
    loads = np.zeros((len(coords), 3)); loads[-1, 1] = P
    d, f, st, neq, kd = bo.solve_model_3dof(np.array(coords), np.array(conn), A, E, np.full(len(conn), I), fix3, loads)
    assert st == 0
    EI, EA = E * I, E * A
    top = d[nc]
    Mc = P * a                                            # constant moment in the column (about z, from the tip load)
    assert top[1] == pytest.approx(P * h / EA, rel=1e-9)                  # axial shortening
    assert top[2] == pytest.approx(Mc * h / EI, rel=1e-9)                 # rotation of the column top
    assert top[0] == pytest.approx(-Mc * h * h / (2 * EI), rel=1e-9)      # sway: a clockwise top rotation moves it to +x for P < 0
    tip = d[-1]
    assert tip[1] == pytest.approx(P * h / EA + top[2] * a + P * a ** 3 / (3 * EI), rel=1e-9)
    assert tip[0] == pytest.approx(top[0], rel=1e-9, abs=1e-15)           # the arm carries no axial force
    # base reactions through the element end forces of the first column element: global (Fx, Fy, Mz) at node 0
    assert f[0, 1] == pytest.approx(-P, rel=1e-9) and f[0, 2] == pytest.approx(-P * a, rel=1e-9) and abs(f[0, 0]) < 1e-6 * abs(P)
