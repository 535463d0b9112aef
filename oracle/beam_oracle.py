"""CPU oracle for the batched Euler-Bernoulli beam FE solve  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic of the reference's hot path lives in the third-party
`openseespy` wheel (un-pinned: /root/reference/environment.yml:13-14, README.md:53),
which is neither vendored under /root/reference nor importable in this image, and the
reference has no tests, golden vectors or data files.  This oracle therefore restates
the published OpenSees semantics selected by the reference's call sites and is pinned
only by (1) closed-form Euler-Bernoulli known answers -- among them the reference's own
six-support bridge by Clapeyron's three-moment equation, and an L-shaped cantilever for the
frame formulation (tests/test_oracle.py) --, (2) two independent formulations in this file
agreeing with each other, (3) equilibrium identities.  tests/test_openseespy_live.py runs
the reference's command sequence against the real module wherever it can be imported.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module.  The product package (`openpystruct_amd/`) never does.

What is restated (reference call sites, all in /root/reference):
  * model build        OpenPyStruct_BeamOpt_training_SingleCore.py:89-124  (setup_model)
  * analyze(1)         ...SingleCore.py:180-182   (BandSPD / RCM / Plain / LoadControl 1.0 / Linear)
  * eleResponse forces ...SingleCore.py:189-190   ([2] = Mz at node I, [1] = Fy at node I)
  * nodeDisp           ...SingleCore.py:224-232   (dof 3 = rotation, dof 2 = deflection)
OpenSees objects whose behaviour is restated: ElasticBeam2d (stiffness, beamUniform
fixed-end forces, getResistingForce), LinearCrdTransf2d (local->global), PlainHandler
(constrained DOFs omitted), BandSPDLinSOE + LAPACK dpbsv.

Two formulations:
  solve_beam_dense      2 DOF/node (u_y, theta_z), constrained DOFs omitted, dense
                        numpy.linalg.solve.
  solve_model_3dof      3 DOF/node (u_x, u_y, theta_z) exactly as OpenSees numbers the
                        reference's model (303 - 2 - 5 = 296 equations, half-bandwidth 5),
                        general 2-D member orientation, axial UDL `Wx` included
                        (SingleCore.py:117 passes the UDL twice), symmetric banded
                        Cholesky via scipy.linalg.solveh_banded (= LAPACK dpbsv, the very
                        routine BandSPD calls).
"""
from __future__ import annotations

import numpy as np

try:  # scipy is present in the image; keep the dense path usable without it
    from scipy.linalg import solveh_banded
except Exception:  # pragma: no cover
    solveh_banded = None


# --------------------------------------------------------------------------------------
# element level (ElasticBeam2d; SingleCore.py:107 `elasticBeamColumn`)
# --------------------------------------------------------------------------------------
def element_stiffness(EI: float, L: float) -> np.ndarray:
    """4x4 Hermite bending stiffness, DOF order (v1, th1, v2, th2)."""
    k = EI / L**3
    return k * np.array(
        [
            [12.0, 6.0 * L, -12.0, 6.0 * L],
            [6.0 * L, 4.0 * L * L, -6.0 * L, 2.0 * L * L],
            [-12.0, -6.0 * L, 12.0, -6.0 * L],
            [6.0 * L, 2.0 * L * L, -6.0 * L, 4.0 * L * L],
        ]
    )


def consistent_udl(w: float, L: float) -> np.ndarray:
    """Consistent nodal loads of `eleLoad -beamUniform Wy` (SingleCore.py:117)."""
    return np.array([w * L / 2.0, w * L * L / 12.0, w * L / 2.0, -w * L * L / 12.0])


def element_end_forces(EI, L, w, v1, t1, v2, t2):
    """Bending part of ElasticBeam2d::getResistingForce for a horizontal member.

    Returns (Fy1, M1, Fy2, M2) = eleResponse(e,'forces')[1], [2], [4], [5]."""
    chord = (v2 - v1) / L
    p1 = t1 - chord
    p2 = t2 - chord
    q1 = 4.0 * EI / L * p1 + 2.0 * EI / L * p2 - w * L * L / 12.0
    q2 = 2.0 * EI / L * p1 + 4.0 * EI / L * p2 + w * L * L / 12.0
    Fy1 = (q1 + q2) / L - w * L / 2.0
    Fy2 = -(q1 + q2) / L - w * L / 2.0
    return Fy1, q1, Fy2, q2


# --------------------------------------------------------------------------------------
# formulation 1: 2 DOF / node, dense
# --------------------------------------------------------------------------------------
def _as_vec(val, n):
    a = np.asarray(val, dtype=np.float64)
    if a.ndim == 0:
        return np.full(n, float(a))
    assert a.shape == (n,), (a.shape, n)
    return a


def assemble_beam(x, E, I, Fy, wy, Mz=None):
    """Global K [2N,2N] and f [2N] before constraints (DOF 2n = u_y, 2n+1 = theta_z)."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    Ne = N - 1
    E = _as_vec(E, Ne)
    I = _as_vec(I, Ne)
    wy = _as_vec(wy, Ne)
    K = np.zeros((2 * N, 2 * N))
    f = np.zeros(2 * N)
    f[0::2] += np.asarray(Fy, dtype=np.float64)
    if Mz is not None:
        f[1::2] += np.asarray(Mz, dtype=np.float64)
    for e in range(Ne):
        L = x[e + 1] - x[e]
        ke = element_stiffness(E[e] * I[e], L)
        d = slice(2 * e, 2 * e + 4)
        K[d, d] += ke
        f[d] += consistent_udl(wy[e], L)
    return K, f


def solve_beam_dense(x, E, I, fix, Fy, wy, Mz=None):
    """One beam.  `fix[n]` bit0 = u_y fixed, bit1 = theta_z fixed (homogeneous SPs,
    SingleCore.py:100-102).  Returns v[N], theta[N], V[Ne], M[Ne], status (0 = ok,
    like `analyze`'s return code, MultiCore.py:182-186)."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    Ne = N - 1
    fix = np.asarray(fix).astype(np.int64)
    K, f = assemble_beam(x, E, I, Fy, wy, Mz)
    free = np.ones(2 * N, dtype=bool)
    free[0::2] = (fix & 1) == 0
    free[1::2] = (fix & 2) == 0
    u = np.zeros(2 * N)
    status = 0
    Kff = K[np.ix_(free, free)]
    try:
        np.linalg.cholesky(Kff)  # SPD check == dpbsv info
        u[free] = np.linalg.solve(Kff, f[free])
    except np.linalg.LinAlgError:
        status = 1
        u[:] = np.nan
    v = u[0::2].copy()
    th = u[1::2].copy()
    Ev = _as_vec(E, Ne)
    Iv = _as_vec(I, Ne)
    wv = _as_vec(wy, Ne)
    V = np.zeros(Ne)
    M = np.zeros(Ne)
    for e in range(Ne):
        L = x[e + 1] - x[e]
        V[e], M[e], _, _ = element_end_forces(Ev[e] * Iv[e], L, wv[e], v[e], th[e], v[e + 1], th[e + 1])
    return v, th, V, M, status


def solve_beam_batched(x, E, I, fix, Fy, wy):
    """Batched convenience wrapper with the C-ABI's broadcasting rules:
    x [N] or [B,N]; E scalar or [B,Ne]; I [B,Ne]; fix [N] or [B,N]; Fy [B,N];
    wy scalar or [B,Ne]."""
    I = np.asarray(I, dtype=np.float64)
    B, Ne = I.shape
    N = Ne + 1
    x = np.asarray(x, dtype=np.float64)
    fix = np.asarray(fix)
    E = np.asarray(E, dtype=np.float64)
    wy = np.asarray(wy, dtype=np.float64)
    Fy = np.asarray(Fy, dtype=np.float64)
    v = np.zeros((B, N))
    th = np.zeros((B, N))
    V = np.zeros((B, Ne))
    M = np.zeros((B, Ne))
    st = np.zeros(B, dtype=np.int32)
    for b in range(B):
        xb = x if x.ndim == 1 else x[b]
        fb = fix if fix.ndim == 1 else fix[b]
        Eb = E if E.ndim == 0 else E[b]
        wb = wy if wy.ndim == 0 else wy[b]
        v[b], th[b], V[b], M[b], st[b] = solve_beam_dense(xb, Eb, I[b], fb, Fy[b], wb)
    return v, th, V, M, st


# --------------------------------------------------------------------------------------
# formulation 2: OpenSees-like 3 DOF / node, banded dpbsv, arbitrary 2-D member orientation
# --------------------------------------------------------------------------------------
def _local_k6(EA, EI, L):
    """6x6 local stiffness of ElasticBeam2d, DOFs (u1, v1, th1, u2, v2, th2)."""
    k = np.zeros((6, 6))
    a = EA / L
    k[0, 0] = k[3, 3] = a
    k[0, 3] = k[3, 0] = -a
    kb = element_stiffness(EI, L)
    idx = [1, 2, 4, 5]
    for r in range(4):
        for c in range(4):
            k[idx[r], idx[c]] = kb[r, c]
    return k


def _rot6(c, s):
    T = np.zeros((6, 6))
    R = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
    T[:3, :3] = R
    T[3:, 3:] = R
    return T


def solve_model_3dof(coords, conn, A, E, I, fix3, nodal_loads, wy=0.0, wx=0.0):
    """General 2-D elastic frame/beam, OpenSees semantics.

    coords [N,2]; conn [Ne,2] 0-based node ids; A, E, I scalars or [Ne];
    fix3 [N,3] 0/1 flags (`ops.fix`); nodal_loads [N,3] (`ops.load`);
    wy, wx scalars or [Ne]: `eleLoad -beamUniform Wy Wx` (local transverse, local axial).

    Returns disp [N,3], forces [Ne,6] (global resisting forces =
    eleResponse(e,'forces')), status, n_eq, half_bandwidth.
    """
    coords = np.asarray(coords, dtype=np.float64)
    conn = np.asarray(conn, dtype=np.int64)
    N = coords.shape[0]
    Ne = conn.shape[0]
    A = _as_vec(A, Ne)
    E = _as_vec(E, Ne)
    I = _as_vec(I, Ne)
    wy = _as_vec(wy, Ne)
    wx = _as_vec(wx, Ne)
    fix3 = np.asarray(fix3).astype(bool)
    # PlainHandler: constrained DOFs get no equation (SingleCore.py:122)
    eq = -np.ones((N, 3), dtype=np.int64)
    n_eq = 0
    for n in range(N):  # node order == RCM order for a chain (SingleCore.py:121)
        for d in range(3):
            if not fix3[n, d]:
                eq[n, d] = n_eq
                n_eq += 1
    f = np.zeros(n_eq)
    for n in range(N):
        for d in range(3):
            if eq[n, d] >= 0:
                f[eq[n, d]] += nodal_loads[n][d]
    # half bandwidth
    kd = 0
    edofs = []
    for e in range(Ne):
        ids = np.concatenate([eq[conn[e, 0]], eq[conn[e, 1]]])
        edofs.append(ids)
        act = ids[ids >= 0]
        if act.size:
            kd = max(kd, int(act.max() - act.min()))
    ab = np.zeros((kd + 1, n_eq))  # upper form: ab[kd + i - j, j] = a[i, j], i <= j
    geo = []
    for e in range(Ne):
        d = coords[conn[e, 1]] - coords[conn[e, 0]]
        L = float(np.hypot(d[0], d[1]))
        c, s = d[0] / L, d[1] / L
        T = _rot6(c, s)
        kl = _local_k6(E[e] * A[e], E[e] * I[e], L)
        kg = T.T @ kl @ T
        # consistent loads = -(resisting force at u = 0) (ElasticBeam2d::addLoad)
        pl = np.array(
            [wx[e] * L / 2, wy[e] * L / 2, wy[e] * L * L / 12, wx[e] * L / 2, wy[e] * L / 2, -wy[e] * L * L / 12]
        )
        pg = T.T @ pl
        ids = edofs[e]
        for r in range(6):
            if ids[r] < 0:
                continue
            f[ids[r]] += pg[r]
            for q in range(6):
                if ids[q] < 0 or ids[q] < ids[r]:
                    continue
                ab[kd + ids[r] - ids[q], ids[q]] += kg[r, q]
        geo.append((L, c, s, T, kl))
    status = 0
    try:
        u = solveh_banded(ab, f, lower=False, check_finite=False)  # LAPACK dpbsv
    except np.linalg.LinAlgError:
        status = 1
        u = np.full(n_eq, np.nan)
    disp = np.zeros((N, 3))
    for n in range(N):
        for d in range(3):
            if eq[n, d] >= 0:
                disp[n, d] = u[eq[n, d]]
    forces = np.zeros((Ne, 6))
    for e in range(Ne):
        L, c, s, T, kl = geo[e]
        ug = np.concatenate([disp[conn[e, 0]], disp[conn[e, 1]]])
        ul = T @ ug
        # basic system (LinearCrdTransf2d / ElasticBeam2d::getResistingForce)
        EA = E[e] * A[e]
        EI = E[e] * I[e]
        chord = (ul[4] - ul[1]) / L
        q0 = EA / L * (ul[3] - ul[0]) - wx[e] * L / 2.0
        q1 = 4 * EI / L * (ul[2] - chord) + 2 * EI / L * (ul[5] - chord) - wy[e] * L * L / 12.0
        q2 = 2 * EI / L * (ul[2] - chord) + 4 * EI / L * (ul[5] - chord) + wy[e] * L * L / 12.0
        p0 = np.array([-wx[e] * L, -wy[e] * L / 2.0, -wy[e] * L / 2.0])
        pl = np.array([-q0 + p0[0], (q1 + q2) / L + p0[1], q1, q0, -(q1 + q2) / L + p0[2], q2])
        forces[e] = T.T @ pl
    return disp, forces, status, n_eq, kd


def solve_reference_beam_3dof(x, A, E, I, roller_nodes, force_nodes, force_values, udl):
    """The exact model `setup_model` builds (SingleCore.py:89-124), 1-based node ids:
    pin at node 1 (fix 1 1 0), rollers (fix 0 1 0), point loads (0, F, 0),
    beamUniform(udl, udl) on every element."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    coords = np.stack([x, np.zeros(N)], axis=1)
    conn = np.stack([np.arange(N - 1), np.arange(1, N)], axis=1)
    fix3 = np.zeros((N, 3), dtype=np.int64)
    fix3[0] = (1, 1, 0)
    for r in roller_nodes:
        fix3[r - 1] = (0, 1, 0)
    loads = np.zeros((N, 3))
    for n, F in zip(force_nodes, force_values):
        loads[n - 1, 1] += F
    return solve_model_3dof(coords, conn, A, E, I, fix3, loads, wy=udl, wx=udl)


# --------------------------------------------------------------------------------------
# the reference's fixed bridge and case distribution (SingleCore.py:20-66, 157-160)
# --------------------------------------------------------------------------------------
E_REF = 200e9
NU_REF = 0.3
G_REF = E_REF / (2 * (1 + NU_REF))
A_REF = 0.01
L_REF = 200.0
N_NODES_REF = 101
ROLLERS_REF = (10, 30, 70, 85, 100)  # 1-based
MAX_FORCE = -355857.0
MIN_FORCE = MAX_FORCE / 10
UDL_REF = -1000.0
I0_REF = 0.5


def reference_fix_mask(num_nodes=N_NODES_REF, rollers=ROLLERS_REF):
    fix = np.zeros(num_nodes, dtype=np.uint8)
    fix[0] = 1
    for r in rollers:
        fix[r - 1] = 1
    return fix


def random_cases(rng, B, num_nodes=N_NODES_REF, rollers=ROLLERS_REF, inertia="uniform"):
    """Synthetic batch with the reference's load distribution (SingleCore.py:157-160)
    and one of SURVEY section 8(d)'s inertia distributions."""
    N = num_nodes
    Ne = N - 1
    avail = np.array([n for n in range(2, N) if n not in rollers])  # 1-based (SingleCore.py:63-66)
    Fy = np.zeros((B, N))
    for b in range(B):
        k = int(rng.integers(1, 5))
        nodes = rng.choice(avail, size=k, replace=False)
        Fy[b, nodes - 1] = rng.uniform(MAX_FORCE, MIN_FORCE, size=k)
    if inertia == "uniform":
        I = np.full((B, Ne), I0_REF)
    elif inertia == "trajectory":
        I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=(B, Ne)))
    elif inertia == "adversarial":
        I = np.exp(rng.uniform(np.log(1e-8), np.log(0.5), size=(B, Ne)))
    else:
        raise ValueError(inertia)
    return I, Fy
