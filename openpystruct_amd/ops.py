"""Per-case façade with the `openseespy.opensees` command names the reference calls.

    import openpystruct_amd.ops as ops        # instead of: import openseespy.opensees as ops

Covers exactly the subset `setup_model` + `generate_sample` use
(/root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py:93-124, :176-190, :224-232):
wipe, model, node, fix, geomTransf, element('elasticBeamColumn'), timeSeries, pattern, load,
eleLoad('-beamUniform'), system, numberer, constraints, integrator, algorithm, analysis, analyze,
eleResponse(e, 'forces'), nodeDisp.  Semantics kept: one process-global domain, 1-based tags,
`analyze` returns 0 on success and a non-zero code (never an exception) when the stiffness matrix is
not positive definite, `eleResponse` returns a fresh list of 6 global end forces
[Fx1, Fy1, Mz1, Fx2, Fy2, Mz2], `nodeDisp(node, dof)` a float (dof 1-based).

The commands only RECORD the model; `analyze(1)` ships it to the GPU as a batch of one through the same
C ABI as the batched operator (there is no CPU path).  For throughput use `deferred()`: models are
queued by `analyze` and solved together by `flush()` in one launch -- the per-case API with the batched
kernel underneath.

Scope: straight beams along x (all nodes on one horizontal line, consecutive tags connected by
consecutive elements: what `setup_model` builds) go to the batched beam kernel; any other 2-D frame of
`elasticBeamColumn`s (what `setup_frame_model` of OpenPyStruct_FrameOpt_Discrete_Beta.py:75-139 builds) goes to
the frame kernel.  Other element types / analyses raise NotImplementedError.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, List, Optional

import numpy as np
import torch

from .beam import beam_solve

_ANALYZE_FAILED = -3   # what OpenSees' StaticAnalysis returns when the solver fails


class _Domain:
    def __init__(self):
        self.ndm = self.ndf = None
        self.nodes: Dict[int, tuple] = {}
        self.fixes: Dict[int, tuple] = {}
        self.elements: Dict[int, tuple] = {}     # tag -> (ni, nj, A, E, Iz)
        self.loads: Dict[int, List[float]] = {}  # node -> [Fx, Fy, Mz]
        self.ele_loads: Dict[int, tuple] = {}    # ele -> (Wy, Wx)
        self.system = None
        self.analysis = None
        self.result = None                       # dict after a successful analyze
        self.device = torch.device("cuda")


_dom = _Domain()
_queue: Optional[List[_Domain]] = None           # deferred mode


class _Stage:
    """Staging of the batch-of-one beam path (one per device and mesh size, reused by every `analyze`): ONE pinned host block and one
    device block for the inputs, one of each for the results, so that a call is two copies in, one launch, two copies out and one stream
    synchronisation -- it used to be six small host-to-device transfers, four device-to-host ones and a sync each (r05,
    scripts/batch_of_one_latency.py: ~500 us per `analyze` of a 100-element beam, of which the launch is 50)."""

    def __init__(self, dev, N):
        Ne = N - 1
        sizes_in = (("x", N), ("E", Ne), ("I", Ne), ("Fy", N), ("wy", Ne))
        sizes_out = (("v", N), ("theta", N), ("V", Ne), ("M", Ne))
        self.off_in, self.off_out, o = {}, {}, 0
        for k, n in sizes_in:                                          # every segment starts 16-byte aligned
            self.off_in[k] = (o, n); o += (n + 1) & ~1
        self.h_in = torch.empty(o, dtype=torch.float64).pin_memory()
        self.d_in = torch.empty(o, dtype=torch.float64, device=dev)
        o = 0
        for k, n in sizes_out:
            self.off_out[k] = (o, n); o += (n + 1) & ~1
        self.h_out = torch.empty(o, dtype=torch.float64).pin_memory()
        self.d_out = torch.empty(o, dtype=torch.float64, device=dev)
        self.h_fix, self.d_fix = torch.empty(N, dtype=torch.uint8).pin_memory(), torch.empty(N, dtype=torch.uint8, device=dev)
        self.h_st, self.d_st = torch.empty(1, dtype=torch.int32).pin_memory(), torch.empty(1, dtype=torch.int32, device=dev)
        self.np_in, self.np_out, self.np_fix = self.h_in.numpy(), self.h_out.numpy(), self.h_fix.numpy()
        dv = lambda buf, off, k, two: buf[off[k][0]:off[k][0] + off[k][1]][None, :] if two else buf[off[k][0]:off[k][0] + off[k][1]]  # noqa: E731
        self.args = dict(x=dv(self.d_in, self.off_in, "x", False), E=dv(self.d_in, self.off_in, "E", True), I=dv(self.d_in, self.off_in, "I", True),
                         fix=self.d_fix, Fy=dv(self.d_in, self.off_in, "Fy", True), wy=dv(self.d_in, self.off_in, "wy", True))
        from .beam import BeamSolution
        self.sol = BeamSolution(*(dv(self.d_out, self.off_out, k, True) for k in ("v", "theta", "V", "M")), self.d_st)

    def solve(self, a, dev):
        for k in ("x", "E", "I", "Fy", "wy"):
            o, n = self.off_in[k]
            self.np_in[o:o + n] = a[k]
        self.np_fix[:] = a["fix"]
        self.d_in.copy_(self.h_in, non_blocking=True)
        self.d_fix.copy_(self.h_fix, non_blocking=True)
        beam_solve(**self.args, out=self.sol)
        self.h_out.copy_(self.d_out, non_blocking=True)
        self.h_st.copy_(self.d_st, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()      # the per-case API hands results back as Python scalars
        res = {k: self.np_out[o:o + n].copy() for k, (o, n) in self.off_out.items()}     # (the staging block is reused by the next call)
        return res, int(self.h_st[0])


_STAGES: Dict[tuple, _Stage] = {}


def wipe():
    global _dom
    dev = _dom.device
    _dom = _Domain()
    _dom.device = dev


def set_device(device):
    _dom.device = torch.device(device)


def model(*args):
    a = list(args)
    if a[:1] != ["basic"] or "-ndm" not in a or "-ndf" not in a:
        raise ValueError("model('basic', '-ndm', 2, '-ndf', 3) expected")
    _dom.ndm, _dom.ndf = int(a[a.index("-ndm") + 1]), int(a[a.index("-ndf") + 1])
    if (_dom.ndm, _dom.ndf) != (2, 3):
        raise NotImplementedError("only -ndm 2 -ndf 3 (the reference's model, SingleCore.py:93)")


def node(tag, x, y):
    _dom.nodes[int(tag)] = (float(x), float(y))


def fix(tag, fx, fy, rz):
    _dom.fixes[int(tag)] = (int(fx), int(fy), int(rz))


def geomTransf(kind, tag, *rest):
    if kind != "Linear":
        raise NotImplementedError("geomTransf 'Linear' only (SingleCore.py:105)")


def element(kind, tag, *args):
    if kind != "elasticBeamColumn":
        raise NotImplementedError("element 'elasticBeamColumn' only (SingleCore.py:107)")
    ni, nj, A, E, Iz = args[:5]     # (ni, nj, A, E, Iz, transfTag)
    _dom.elements[int(tag)] = (int(ni), int(nj), float(A), float(E), float(Iz))


def timeSeries(kind, tag, *rest):
    if kind != "Linear":
        raise NotImplementedError("timeSeries 'Linear' only")


def pattern(kind, tag, ts_tag, *rest):
    if kind != "Plain":
        raise NotImplementedError("pattern 'Plain' only")


def load(node_tag, Fx, Fy, Mz):
    v = _dom.loads.setdefault(int(node_tag), [0.0, 0.0, 0.0])
    v[0] += float(Fx); v[1] += float(Fy); v[2] += float(Mz)


def eleLoad(*args):
    a = list(args)
    if "-ele" not in a or "-type" not in a or "-beamUniform" not in a:
        raise NotImplementedError("eleLoad('-ele', e, '-type', '-beamUniform', Wy, <Wx>) only (SingleCore.py:117)")
    e = int(a[a.index("-ele") + 1])
    vals = [float(t) for t in a[a.index("-beamUniform") + 1:]]
    wy = vals[0]
    wx = vals[1] if len(vals) > 1 else 0.0      # the SECOND value is the axial UDL
    old = _dom.ele_loads.get(e, (0.0, 0.0))
    _dom.ele_loads[e] = (old[0] + wy, old[1] + wx)


def system(kind, *rest):
    """Every system of equations is solved by the same positive-pivot band LDL^T (the kernels' only factorisation): exact for
    the SPD stiffness matrices of the reference's models under ANY of these names.  A general-matrix solver (`BandGeneral` =
    LAPACK dgbsv, FR:134) would also get through an indefinite but non-singular matrix -- a mechanism held by a negative
    stiffness, which the reference's commands cannot build; here `analyze` reports failure for it (non-zero return, like a
    singular matrix)."""
    if kind not in ("BandSPD", "BandGeneral", "ProfileSPD", "FullGeneral", "UmfPack", "SparseGeneral"):
        raise ValueError(kind)
    _dom.system = kind


# analysis-object commands: only what is equivalent to ONE linear static solve with constrained DOFs dropped is accepted
# (SingleCore.py:121-124, FR:135-138); anything else would silently change the meaning of analyze(1), so it raises
def numberer(kind="RCM", *a):
    if kind not in ("RCM", "Plain", "AMD"):          # a numbering only permutes equations: results are identical
        raise NotImplementedError(f"numberer {kind!r}")


def constraints(kind="Plain", *a):
    if kind not in ("Plain", "Transformation"):      # homogeneous single-point constraints: both drop the constrained DOFs
        raise NotImplementedError(f"constraints {kind!r}: only homogeneous fixities are modelled (SingleCore.py:100-102, :122)")


def integrator(kind="LoadControl", *a):
    if kind != "LoadControl" or (a and float(a[0]) != 1.0):
        raise NotImplementedError(f"integrator {kind!r} {a!r}: one load step to load factor 1.0 (SingleCore.py:123)")


def algorithm(kind="Linear", *a):
    if kind not in ("Linear", "Newton", "ModifiedNewton"):   # a linear problem: Newton converges in its first iteration (FR:138)
        raise NotImplementedError(f"algorithm {kind!r}")


def analysis(kind):
    if kind != "Static":
        raise NotImplementedError("analysis 'Static' only")
    _dom.analysis = kind


def _is_straight_beam(d: _Domain) -> bool:
    tags = sorted(d.nodes)
    N = len(tags)
    if N < 2 or tags != list(range(1, N + 1)):
        return False
    xy = np.array([d.nodes[t] for t in tags])
    if np.any(xy[:, 1] != xy[0, 1]) or np.any(np.diff(xy[:, 0]) <= 0):
        return False
    return sorted(d.elements) == list(range(1, N)) and all(d.elements[e][:2] == (e, e + 1) for e in d.elements)


def _frame_arrays(d: _Domain):
    """The recorded domain as the arrays of a general 2-D frame (what `setup_frame_model` builds,
    OpenPyStruct_FrameOpt_Discrete_Beta.py:75-139)."""
    tags = sorted(d.nodes)
    N = len(tags)
    if tags != list(range(1, N + 1)) or sorted(d.elements) != list(range(1, len(d.elements) + 1)):
        raise NotImplementedError("node and element tags must be 1..N / 1..Ne")
    coords = np.array([d.nodes[t] for t in tags])
    Ne = len(d.elements)
    conn = np.array([[d.elements[e][0] - 1, d.elements[e][1] - 1] for e in range(1, Ne + 1)])
    A = np.array([d.elements[e][2] for e in range(1, Ne + 1)])
    E = np.array([d.elements[e][3] for e in range(1, Ne + 1)])
    I = np.array([d.elements[e][4] for e in range(1, Ne + 1)])
    fix3 = np.zeros((N, 3), dtype=bool)
    for t, f in d.fixes.items():
        fix3[t - 1] = [bool(v) for v in f]
    loads = np.zeros((N, 3))
    for t, f in d.loads.items():
        loads[t - 1] = f
    wy = np.array([d.ele_loads.get(e, (0.0, 0.0))[0] for e in range(1, Ne + 1)])
    wx = np.array([d.ele_loads.get(e, (0.0, 0.0))[1] for e in range(1, Ne + 1)])
    return dict(coords=coords, conn=conn, fix3=fix3, A=A, E=E, wy=wy, wx=wx, loads=loads, I=I)


def _frame_finish(d: _Domain, disp, forces, status):
    if status != 0:
        d.result = None
        return _ANALYZE_FAILED
    d.result = dict(ux=disp[:, 0], v=disp[:, 1], th=disp[:, 2], forces=forces)
    return 0


_TOPO_CACHE: "dict" = {}       # the last few topologies: setup_frame_model rebuilds the SAME model every epoch (FR:178-183)


def _topology(a0, device):
    """FrameTopology of a recorded frame, cached on its geometry / connectivity / constraints / sections / element loads: the equation
    numbering (r05: reverse Cuthill-McKee and sweeps, 6-12 ms of host time at 10 x 10 / 15 x 16), the device tables and the factor
    workspace are built once per model, not once per `analyze`."""
    from .frames import FrameTopology
    key = (str(torch.device(device)),) + tuple(np.ascontiguousarray(a0[k]).tobytes() for k in ("coords", "conn", "fix3", "A", "E", "wy", "wx"))
    topo = _TOPO_CACHE.pop(key, None)
    if topo is None:
        topo = FrameTopology(a0["coords"], a0["conn"], a0["fix3"], a0["A"], a0["E"], a0["wy"], a0["wx"], a0["loads"], device=device)
    _TOPO_CACHE[key] = topo                          # (re-inserted: most recently used last)
    while len(_TOPO_CACHE) > 8:
        _TOPO_CACHE.pop(next(iter(_TOPO_CACHE)))
    return topo


def _solve_frames(arrs, device):
    """Frames of ONE topology (same nodes, connectivity, fixities, sections, element loads), different inertias and nodal
    loads: one launch of the batched frame kernel."""
    from .frames import frame_solve
    a0 = arrs[0]
    topo = _topology(a0, device)
    I = torch.as_tensor(np.stack([a["I"] for a in arrs]), dtype=torch.float64, device=device)
    loads = torch.as_tensor(np.stack([a["loads"] for a in arrs]), dtype=torch.float64, device=device)
    sol = frame_solve(topo, I, loads=loads)
    return sol.disp.cpu().numpy(), sol.forces.cpu().numpy(), sol.status.cpu().numpy()


def _device_of(d: _Domain):
    if not torch.cuda.is_available():
        raise RuntimeError("openpystruct_amd.ops: analyze() needs a GPU -- the library has no CPU fallback")
    dev = torch.device(d.device)
    return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())


class _FrameStage:
    """As `_Stage`, for the batch-of-one frame path; kept on the cached topology (the script rebuilds the same frame every epoch)."""

    def __init__(self, topo, dev):
        from .frames import FrameSolution
        Nn, Ne = topo.Nn, topo.Ne
        r2 = lambda n: (n + 1) & ~1  # noqa: E731
        self.oI, self.oL = 0, r2(Ne)
        n_in = self.oL + 3 * Nn
        self.h_in, self.d_in = torch.empty(n_in, dtype=torch.float64).pin_memory(), torch.empty(n_in, dtype=torch.float64, device=dev)
        self.oD, self.oF = 0, r2(3 * Nn)
        oV = self.oF + 6 * Ne
        oM = oV + r2(Ne)
        n_out = oM + Ne
        self.h_out, self.d_out = torch.empty(n_out, dtype=torch.float64).pin_memory(), torch.empty(n_out, dtype=torch.float64, device=dev)
        self.h_st, self.d_st = torch.empty(1, dtype=torch.int32).pin_memory(), torch.empty(1, dtype=torch.int32, device=dev)
        self.np_in, self.np_out = self.h_in.numpy(), self.h_out.numpy()
        self.I = self.d_in[:Ne][None, :]
        self.loads = self.d_in[self.oL:self.oL + 3 * Nn].view(1, Nn, 3)
        self.sol = FrameSolution(self.d_out[:3 * Nn].view(1, Nn, 3), self.d_out[self.oF:self.oF + 6 * Ne].view(1, Ne, 6),
                                 self.d_out[oV:oV + Ne][None, :], self.d_out[oM:oM + Ne][None, :], self.d_st)
        self.Nn, self.Ne = Nn, Ne

    def solve(self, topo, a, dev):
        from .frames import frame_solve
        self.np_in[:self.Ne] = a["I"]
        self.np_in[self.oL:self.oL + 3 * self.Nn] = a["loads"].reshape(-1)
        self.d_in.copy_(self.h_in, non_blocking=True)
        frame_solve(topo, self.I, loads=self.loads, out=self.sol)
        self.h_out.copy_(self.d_out, non_blocking=True)
        self.h_st.copy_(self.d_st, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        disp = self.np_out[:3 * self.Nn].reshape(self.Nn, 3).copy()
        forces = self.np_out[self.oF:self.oF + 6 * self.Ne].reshape(self.Ne, 6).copy()
        return disp, forces, int(self.h_st[0])


def _analyze_frame(d: _Domain):
    """General 2-D frame: the batched frame kernel with a batch of one."""
    a = _frame_arrays(d)
    dev = _device_of(d)
    topo = _topology(a, dev)
    stage = topo.__dict__.get("_shim_stage")
    if stage is None:
        stage = topo.__dict__["_shim_stage"] = _FrameStage(topo, dev)
    with torch.cuda.device(dev):
        disp, forces, status = stage.solve(topo, a, dev)
    return _frame_finish(d, disp, forces, status)


def _arrays(d: _Domain):
    tags = sorted(d.nodes)
    N = len(tags)
    if N < 2 or tags != list(range(1, N + 1)):
        raise NotImplementedError("node tags must be 1..N")
    xy = np.array([d.nodes[t] for t in tags])
    if np.any(xy[:, 1] != xy[0, 1]) or np.any(np.diff(xy[:, 0]) <= 0):
        raise NotImplementedError("straight beam along +x only (what setup_model builds)")
    Ne = N - 1
    if sorted(d.elements) != list(range(1, Ne + 1)) or any(d.elements[e][:2] != (e, e + 1) for e in d.elements):
        raise NotImplementedError("element e must connect nodes e and e+1")
    E = np.array([d.elements[e][3] for e in range(1, Ne + 1)])
    A = np.array([d.elements[e][2] for e in range(1, Ne + 1)])
    I = np.array([d.elements[e][4] for e in range(1, Ne + 1)])
    fixb = np.zeros(N, dtype=np.uint8)
    fx = np.zeros(N, dtype=bool)
    for t, (a, b, c) in d.fixes.items():
        fixb[t - 1] = (1 if b else 0) | (2 if c else 0)
        fx[t - 1] = bool(a)
    Fy = np.zeros(N); Fx = np.zeros(N); Mz = np.zeros(N)
    for t, (a, b, c) in d.loads.items():
        Fx[t - 1], Fy[t - 1], Mz[t - 1] = a, b, c
    if np.any(Mz != 0.0):
        raise NotImplementedError("nodal moments are not on the reference's path (ops.load(node, 0, F, 0), SingleCore.py:113)")
    wy = np.array([d.ele_loads.get(e, (0.0, 0.0))[0] for e in range(1, Ne + 1)])
    wx = np.array([d.ele_loads.get(e, (0.0, 0.0))[1] for e in range(1, Ne + 1)])
    return dict(x=xy[:, 0], E=E, A=A, I=I, fix=fixb, fix_x=fx, Fy=Fy, Fx=Fx, wy=wy, wx=wx)


def _axial(a, N):
    """Axial sub-problem on the host: it decouples on a straight beam and the reference never reads it
    (SURVEY fact 9); kept so that eleResponse(...)[0], [3] and nodeDisp(n, 1) are OpenSees-like.
    Supported: exactly one node restrained in x (the pin of `setup_model`)."""
    x, L = a["x"], np.diff(a["x"])
    fixed = np.nonzero(a["fix_x"])[0]
    if fixed.size != 1:
        return None, None
    k = int(fixed[0])
    Ne = N - 1
    # tension T just right of node n / just left: integrate applied axial load from the free ends (running sums: this runs once per
    # `analyze`, i.e. once per epoch of the reference's loop -- a Python loop over the elements was a fifth of the call)
    q = a["wx"] * L                                   # total axial load per element
    Fx = a["Fx"]
    e = np.arange(Ne)
    suf_q = np.concatenate([np.cumsum(q[::-1])[::-1], [0.0]])        # suf_q[i] = q[i:].sum()
    suf_F = np.concatenate([np.cumsum(Fx[::-1])[::-1], [0.0]])       # suf_F[i] = Fx[i:].sum()
    pre_q = np.concatenate([[0.0], np.cumsum(q)])                    # pre_q[i] = q[:i].sum()
    pre_F = np.cumsum(Fx)                                            # pre_F[i] = Fx[:i + 1].sum()
    right = e >= k       # right of the support: everything to the right hangs on this section; left of it: everything to the left
    T_right = np.where(right, suf_q[e + 1] + suf_F[e + 1], -(pre_q[e] + pre_F[e]) - q)
    T_left = np.where(right, T_right + q, -(pre_q[e] + pre_F[e]))
    dux = 0.5 * (T_left + T_right) / (a["E"] * a["A"]) * L          # elongation per element
    ux = np.zeros(N)
    ux[k + 1:] = np.cumsum(dux[k:])
    if k > 0:
        ux[:k] = -np.cumsum(dux[:k][::-1])[::-1]
    return (T_left, T_right), ux


def _finish(d: _Domain, a, v, th, V, M, status):
    if status != 0:
        d.result = None
        return _ANALYZE_FAILED
    N = len(v)
    L = np.diff(a["x"])
    Fy2 = -V - a["wy"] * L
    M2 = (V + 0.5 * a["wy"] * L) * L - M
    T, ux = _axial(a, N)
    Fx1 = -T[0] if T is not None else np.zeros(N - 1)
    Fx2 = T[1] if T is not None else np.zeros(N - 1)
    d.result = dict(v=v, th=th, ux=ux if ux is not None else np.zeros(N),
                    forces=np.stack([Fx1, V, M, Fx2, Fy2, M2], axis=1))
    return 0


def analyze(n_steps=1):
    """One linear static step (SingleCore.py:182).  Returns 0, or a non-zero code if the system is not SPD."""
    d = _dom
    if d.analysis != "Static":
        return _ANALYZE_FAILED
    if not _is_straight_beam(d):
        if _queue is not None:          # deferred: frames of equal topology are solved in one launch at flush
            d._frame = _frame_arrays(d)
            _queue.append(d)
            return 0
        return _analyze_frame(d)
    a = _arrays(d)
    if _queue is not None:
        d._arrays = a
        _queue.append(d)
        return 0
    dev = _device_of(d)
    N = len(a["x"])
    stage = _STAGES.get((dev, N))
    if stage is None:
        if len(_STAGES) >= 8:
            _STAGES.clear()
        stage = _STAGES[(dev, N)] = _Stage(dev, N)
    with torch.cuda.device(dev):
        r, st = stage.solve(a, dev)
    return _finish(d, a, r["v"], r["theta"], r["V"], r["M"], st)


def eleResponse(ele_tag, what):
    if what != "forces":
        raise NotImplementedError("eleResponse(e, 'forces') only (SingleCore.py:189-190)")
    if _dom.result is None:
        raise RuntimeError("no committed state: analyze() has not succeeded")
    return [float(f) for f in _dom.result["forces"][int(ele_tag) - 1]]


def nodeDisp(node_tag, dof):
    if _dom.result is None:
        raise RuntimeError("no committed state: analyze() has not succeeded")
    key = {1: "ux", 2: "v", 3: "th"}[int(dof)]
    return float(_dom.result[key][int(node_tag) - 1])


@contextmanager
def deferred(device=None):
    """Queue the models that `analyze` is called on and solve them in ONE launch at `flush()` / exit.

        with ops.deferred() as batch:
            for case in cases:
                ops.wipe(); setup_model(...); ops.analysis('Static'); ops.analyze(1)
        results = batch.domains          # each has .result like the global domain after analyze

    Queued straight beams must have the same number of nodes; queued frames are grouped by topology (one launch per group,
    inertias and nodal loads per frame)."""
    global _queue

    class _Batch:
        domains: List[_Domain] = []
        codes: List[int] = []

    batch = _Batch()
    _queue = []
    try:
        yield batch
    finally:
        q, _queue = _queue, None
        if q:
            dev = torch.device(device) if device is not None else q[0].device
            batch.domains = q
            codes = [None] * len(q)
            beams = [i for i, d in enumerate(q) if getattr(d, "_frame", None) is None]
            frames_ = [i for i, d in enumerate(q) if getattr(d, "_frame", None) is not None]
            if beams:
                arrs = [q[i]._arrays for i in beams]
                N = len(arrs[0]["x"])
                if any(len(a["x"]) != N for a in arrs):
                    raise NotImplementedError("deferred(): all beam models must have the same number of nodes")
                st = lambda k: np.stack([a[k] for a in arrs])  # noqa: E731
                t = lambda z, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(z), dtype=dt, device=dev)  # noqa: E731
                sol = beam_solve(t(st("x")), t(st("E")), t(st("I")), t(st("fix"), torch.uint8), t(st("Fy")), t(st("wy")))
                v, th, V, M = (z.cpu().numpy() for z in (sol.v, sol.theta, sol.V, sol.M))
                status = sol.status.cpu().numpy()
                for k, i in enumerate(beams):
                    codes[i] = _finish(q[i], arrs[k], v[k], th[k], V[k], M[k], int(status[k]))
            # frames: one launch per distinct topology (the frame script rebuilds the SAME frame every epoch, FR:178-183)
            groups = {}
            for i in frames_:
                f = q[i]._frame
                key = (f["coords"].tobytes(), f["conn"].tobytes(), f["fix3"].tobytes(), f["A"].tobytes(), f["E"].tobytes(),
                       f["wy"].tobytes(), f["wx"].tobytes())
                groups.setdefault(key, []).append(i)
            for idxs in groups.values():
                disp, forces, status = _solve_frames([q[i]._frame for i in idxs], dev)
                for k, i in enumerate(idxs):
                    codes[i] = _frame_finish(q[i], disp[k], forces[k], int(status[k]))
            batch.codes = codes
