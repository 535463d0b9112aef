"""Batched 2-D frame analysis and sizing: the host side of SURVEY 8(f1) / BASELINE config 5.

Mirrors /root/reference/OpenPyStruct_FrameOpt_Discrete_Beta.py:
  * grid geometry, element order (columns first, then beams), supports, loads   FR:50-69, :75-139 -> `grid_frame`
  * `setup_frame_model` + `ops.analyze(1)` + `ops.eleResponse(e,'forces')`        FR:75-139, :151, :181-183 -> `frame_solve`
  * loss (bending eps 1e-8, "shear" = global Fy even for columns), Adam without a scheduler, early stop
                                                                                  FR:141-206 -> `optimize_frames`
One topology (coordinates, connectivity, constraints) is prepared once on the host -- equation numbers as
OpenSees' PlainHandler + a node-order numberer would give them, half bandwidth -- and shared by the batch;
frames differ in their inertia vectors (and optionally loads).  The solve is the HIP kernel in
csrc/frame_solve.hip behind `ops_frame_solve_batched_f64`; there is no CPU path.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _cabi


@dataclass
class FrameConfig:
    """FR:17-44."""
    bay_width: float = 6.0
    story_height: float = 3.0
    E: float = 200e9
    nu: float = 0.3
    A: float = 0.02
    I0: float = 5e-4
    alpha_moment: float = 1e-2
    alpha_shear: float = 1e-2
    k: float = 0.03
    lateral_load: float = 1e4
    vertical_load: float = -1e4
    num_epochs: int = 5000
    lr: float = 0.005
    tolerance: float = 1e-3
    patience: int = 10

    @property
    def G(self):
        return self.E / (2 * (1 + self.nu))


def rcm_node_order(n_nodes: int, conn: np.ndarray, has_eq: np.ndarray) -> np.ndarray:
    """Reverse Cuthill-McKee order of the nodes that carry equations (what `ops.numberer('RCM')`, FR:135 / SC:121, asks OpenSees
    for): breadth-first levels from a pseudo-peripheral node (George-Liu: repeat from a minimum-degree node of the last level
    while the eccentricity grows), neighbours by increasing degree, the whole order reversed; components one after the other.
    Nodes without equations (fully fixed) keep their place at the end -- they number nothing."""
    adj = [set() for _ in range(n_nodes)]
    for a, b in conn:
        a, b = int(a), int(b)
        if a != b and has_eq[a] and has_eq[b]:
            adj[a].add(b); adj[b].add(a)
    deg = np.array([len(s) for s in adj])
    todo = set(int(i) for i in np.nonzero(has_eq)[0])

    def levels(root):
        seen, order, frontier, depth = {root}, [root], [root], 0
        while frontier:
            nxt = []
            for u in frontier:
                for v in sorted(adj[u] - seen, key=lambda w: (deg[w], w)):
                    seen.add(v); nxt.append(v)
            if not nxt:
                break
            order += nxt; frontier = nxt; depth += 1
        return order, frontier, depth

    out = []
    while todo:
        root = min(todo, key=lambda w: (deg[w], w))
        order, last, depth = levels(root)
        while True:
            cand = min(last, key=lambda w: (deg[w], w))
            o2, l2, d2 = levels(cand)
            if d2 <= depth:
                break
            root, order, last, depth = cand, o2, l2, d2
        out += order[::-1]
        todo -= set(order)
    return np.array(out + [int(i) for i in np.nonzero(~has_eq)[0]], dtype=np.int64)


def _number_equations(order: np.ndarray, fix3: np.ndarray, conn: np.ndarray):
    """Equation numbers node by node along `order` (constrained DOFs get none: constraints('Plain')), and the half bandwidth."""
    node_eq = -np.ones(fix3.shape, dtype=np.int32)
    k = 0
    for nd in order:
        for dof in range(3):
            if not fix3[nd, dof]:
                node_eq[nd, dof] = k
                k += 1
    elem_eq = np.concatenate([node_eq[conn[:, 0]], node_eq[conn[:, 1]]], axis=1)
    span = [int(q[q >= 0].max() - q[q >= 0].min()) for q in elem_eq if (q >= 0).any()]
    return node_eq, elem_eq, (max(span) if span else 0)


class FrameTopology:
    """Shared description of a frame: what `setup_frame_model` rebuilds every epoch, minus the inertias.

    `numbering`: "node" = equations in node order (what r01-r04 did: row by row on a grid); "rcm" = reverse Cuthill-McKee (the
    reference's `numberer('RCM')`, FR:135); "auto" (default) = the narrowest of node order, reverse Cuthill-McKee and the two
    coordinate sweeps (`self.numbering` says which), node order on a tie.  The solution does not depend on it beyond rounding; the work does (n kd^2): a 10-bay x 2-story frame of the reference's own
    random range is kd 35 story by story and kd 8 along its column lines, and a 21 x 3 frame (kd 68 in node order: beyond the
    tuned kernels' 63, i.e. on the slow column-by-column fallback) stays on the wave kernel."""

    def __init__(self, coords, conn, fix3, A, E, wy, wx, nodal_loads, device="cuda", numbering: str = "auto"):
        coords = np.asarray(coords, dtype=np.float64)
        conn = np.asarray(conn, dtype=np.int64)
        fix3 = np.asarray(fix3).astype(bool)
        self.Nn, self.Ne = coords.shape[0], conn.shape[0]
        d = coords[conn[:, 1]] - coords[conn[:, 0]]
        L = np.hypot(d[:, 0], d[:, 1])
        geo = np.stack([L, d[:, 0] / L, d[:, 1] / L], axis=1)
        if numbering not in ("auto", "node", "rcm"):
            raise ValueError("numbering must be 'auto', 'node' or 'rcm'")
        node_eq, elem_eq, kd = _number_equations(np.arange(self.Nn), fix3, conn)
        self.numbering = "node"
        if numbering == "rcm":
            node_eq, elem_eq, kd = _number_equations(rcm_node_order(self.Nn, conn, ~fix3.all(axis=1)), fix3, conn)
            self.numbering = "rcm"
        elif numbering == "auto":
            # candidates: reverse Cuthill-McKee, and the two coordinate sweeps (nodes sorted by (x, y) / by (y, x): on a rectangular grid
            # the sweep along the shorter side is the optimum, which the breadth-first levels of RCM -- diagonals of the grid -- miss)
            cands = [("rcm", rcm_node_order(self.Nn, conn, ~fix3.all(axis=1))),
                     ("sweep-x", np.lexsort((coords[:, 1], coords[:, 0]))), ("sweep-y", np.lexsort((coords[:, 0], coords[:, 1])))]
            for name, order in cands:
                c_node_eq, c_elem_eq, c_kd = _number_equations(order, fix3, conn)
                if c_kd < kd:
                    node_eq, elem_eq, kd, self.numbering = c_node_eq, c_elem_eq, c_kd, name
        self.n_eq, self.kd = int((~fix3).sum()), kd
        Ev = np.broadcast_to(np.asarray(E, dtype=np.float64), (self.Ne,))
        Av = np.broadcast_to(np.asarray(A, dtype=np.float64), (self.Ne,))
        w = np.stack([np.broadcast_to(np.asarray(wy, dtype=np.float64), (self.Ne,)),
                      np.broadcast_to(np.asarray(wx, dtype=np.float64), (self.Ne,))], axis=1)
        self.coords, self.conn, self.fix3 = coords, conn, fix3
        self.A, self.E, self.wy, self.wx = Av.copy(), Ev.copy(), w[:, 0].copy(), w[:, 1].copy()
        self.nodal_loads = np.asarray(nodal_loads, dtype=np.float64).reshape(self.Nn, 3)
        dev = torch.device(device)
        t = lambda a, dt: torch.as_tensor(np.array(a), dtype=dt, device=dev)  # noqa: E731
        self.device = dev
        self.d_geo, self.d_EA, self.d_E, self.d_w = t(geo, torch.float64), t(Ev * Av, torch.float64), t(Ev, torch.float64), t(w, torch.float64)
        self.d_elem_eq, self.d_node_eq = t(elem_eq, torch.int32), t(node_eq, torch.int32)
        self.d_loads = t(self.nodal_loads, torch.float64)

    def lds_bytes(self) -> int:
        n3, ld = (self.n_eq + 2) // 3 * 3, (max(self.kd, 3) + 4) & ~1     # csrc/frame_solve.hip: frame_n3, frame_ld
        return (n3 * ld + n3) * 8


def grid_frame(num_bays: int, num_stories: int, cfg: Optional[FrameConfig] = None, device="cuda", numbering: str = "auto") -> FrameTopology:
    """The reference's rectangular frame (FR:50-69, :84-131): nodes row by row from the ground, columns then
    beams, ground row fully fixed, lateral loads on the left column line, beamUniform(w, w) on the beams."""
    cfg = cfg or FrameConfig()
    nb1 = num_bays + 1
    coords = np.array([(j * cfg.bay_width, i * cfg.story_height) for i in range(num_stories + 1) for j in range(nb1)])
    cols = [(i * nb1 + j, (i + 1) * nb1 + j) for i in range(num_stories) for j in range(nb1)]                  # FR:101-107
    beams = [(i * nb1 + j, i * nb1 + j + 1) for i in range(1, num_stories + 1) for j in range(num_bays)]      # FR:110-116
    conn = np.array(cols + beams)
    fix3 = np.zeros((coords.shape[0], 3), dtype=bool)
    fix3[coords[:, 1] == 0.0] = True                                                                           # FR:96-98
    loads = np.zeros((coords.shape[0], 3))
    loads[(coords[:, 0] == 0.0) & (coords[:, 1] != 0.0), 0] = cfg.lateral_load                                 # FR:126-128
    w = np.zeros(len(conn)); w[len(cols):] = cfg.vertical_load                                                 # FR:130-131 (Wy = Wx)
    return FrameTopology(coords, conn, fix3, cfg.A, cfg.E, w, w, loads, device, numbering=numbering)


class FrameSolution(NamedTuple):
    disp: torch.Tensor      # [B, Nn, 3]
    forces: torch.Tensor    # [B, Ne, 6]  eleResponse(e, 'forces')
    V: torch.Tensor         # [B, Ne]     forces[..., 1]  (FR:152)
    M: torch.Tensor         # [B, Ne]     forces[..., 2]  (FR:153)
    status: torch.Tensor    # [B] int32


def frame_solve(topo: FrameTopology, I: torch.Tensor, loads: Optional[torch.Tensor] = None,
                out: Optional[FrameSolution] = None) -> FrameSolution:
    lib = _cabi.load()
    if not torch.is_tensor(I) or not I.is_cuda:
        raise RuntimeError("frame_solve needs GPU tensors: openpystruct_amd has no CPU fallback")
    if I.dtype != torch.float64 or I.dim() != 2 or I.shape[1] != topo.Ne:
        raise ValueError(f"I must be float64 [B, {topo.Ne}]")
    I = I.contiguous()
    B = I.shape[0]
    dev = I.device
    if loads is None:
        loads, lbs = topo.d_loads, 0
    else:
        loads = loads.to(torch.float64).contiguous()
        lbs = topo.Nn * 3 if loads.dim() == 3 else 0
    if out is None:
        f64 = dict(dtype=torch.float64, device=dev)
        out = FrameSolution(torch.empty((B, topo.Nn, 3), **f64), torch.empty((B, topo.Ne, 6), **f64),
                            torch.empty((B, topo.Ne), **f64), torch.empty((B, topo.Ne), **f64),
                            torch.empty((B,), dtype=torch.int32, device=dev))
    ws_bytes = int(lib.ops_frame_workspace_bytes(B, topo.n_eq, topo.kd))
    ws, flags, entry = None, 0, None
    if ws_bytes:       # factor storage + the topology's assembly plan: HBM workspace, cached on the topology PER STREAM (two solves on one
        # topology from different streams or threads must not share factor columns or plan)
        cache = topo.__dict__.setdefault("_ws", {})
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        entry = cache.get(key)
        if entry is None or entry[0].numel() < ws_bytes:
            entry = cache[key] = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev), 0]
        ws = entry[0]
        # the plan (the topology-only part of the assembly, at the start of the workspace) is built by the first call that uses this buffer and
        # kept: a FrameTopology's arrays never change (include/openpystruct_amd.h OPS_FRAME_REUSE_PLAN)
        sig = int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd))
        if sig != 0 and entry[1] == sig:
            flags = _cabi.FRAME_REUSE_PLAN
    with torch.cuda.device(dev):
        rc = lib.ops_frame_solve_batched_f64_ex(
            B, topo.Nn, topo.Ne, topo.n_eq, topo.kd, topo.d_geo.data_ptr(), topo.d_EA.data_ptr(), topo.d_E.data_ptr(),
            topo.d_w.data_ptr(), topo.d_elem_eq.data_ptr(), topo.d_node_eq.data_ptr(), I.data_ptr(), loads.data_ptr(), lbs,
            out.disp.data_ptr(), out.forces.data_ptr(), out.V.data_ptr(), out.M.data_ptr(), out.status.data_ptr(),
            ws.data_ptr() if ws is not None else None, ws_bytes, torch.cuda.current_stream(dev).cuda_stream, flags)
    if entry is not None:
        entry[1] = sig if rc == _cabi.OK else 0
    if rc == _cabi.ERR_UNSUPPORTED:
        raise NotImplementedError(f"frame too large: n_eq={topo.n_eq}, half bandwidth={topo.kd} (half bandwidth <= 63: a (kd+6)-column ring, one "
                                  f"n_eq vector and two 24-column chunks must fit 160 KB of LDS; beyond 63, up to 1024: one n_eq vector and one column)")
    if rc != _cabi.OK:
        raise RuntimeError(f"ops_frame_solve_batched_f64 failed with code {rc}")
    return out


def optimize_frames(topo: FrameTopology, B: int, cfg: Optional[FrameConfig] = None, I0: Optional[torch.Tensor] = None,
                    max_epochs: Optional[int] = None, poll_every: int = 25, loss_history: Optional[list] = None):
    """FR:163-206 for B frames at once (same topology; `I0` [B,Ne] lets them start from different designs).
    Adam(lr) with NO scheduler (gamma = 1), loss with `+1e-8` in the bending term (FR:155), early stop
    tolerance 1e-3 / patience 10.  Returns (I float32 [B,Ne], solution of the last solve, epochs_run).
    `loss_history`: a list that receives every epoch's `total_loss` [B] (FR:190; a stopped frame repeats its last value)."""
    cfg = cfg or FrameConfig()
    lib = _cabi.load()
    dev = topo.device
    Ne = topo.Ne
    f32 = dict(dtype=torch.float32, device=dev)
    I = (I0.to(**f32).clone() if I0 is not None else torch.full((B, Ne), cfg.I0, **f32))
    I64 = I.double()
    ea, es = torch.zeros((B, Ne), **f32), torch.zeros((B, Ne), **f32)
    best = torch.full((B,), float("inf"), **f32)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    ep = torch.zeros((B,), dtype=torch.int32, device=dev)
    active = torch.ones((B,), dtype=torch.uint8, device=dev)
    last = torch.zeros((B,), **f32)
    V32, M32 = torch.zeros((B, Ne), **f32), torch.zeros((B, Ne), **f32)
    n_max = max_epochs if max_epochs is not None else cfg.num_epochs
    hp = _cabi.SizingParams(E=cfg.E, G=cfg.G, alpha_moment=cfg.alpha_moment, alpha_shear=cfg.alpha_shear, lr=cfg.lr, gamma=1.0,
                            beta1=0.9, beta2=0.999, adam_eps=1e-8, clamp_min=1e-8, bend_eps=1e-8, area_coef=cfg.k,
                            tolerance=cfg.tolerance, patience=cfg.patience, max_epochs=n_max)
    sol = None
    for e in range(n_max):
        sol = frame_solve(topo, I64, out=sol)
        with torch.cuda.device(dev):
            rc = lib.ops_beam_sizing_step_f32(B, Ne, I.data_ptr(), I64.data_ptr(), sol.V.data_ptr(), sol.M.data_ptr(), ea.data_ptr(),
                                              es.data_ptr(), best.data_ptr(), cnt.data_ptr(), ep.data_ptr(), active.data_ptr(),
                                              last.data_ptr(), V32.data_ptr(), M32.data_ptr(), ctypes.byref(hp),
                                              torch.cuda.current_stream(dev).cuda_stream)
        if rc != _cabi.OK:
            raise RuntimeError(f"ops_beam_sizing_step_f32 failed with code {rc}")
        if loss_history is not None:
            loss_history.append(last.clone())
        if (e + 1) % poll_every == 0 and not bool(active.any()):
            break
    torch.cuda.synchronize(dev)
    return I, sol, ep
