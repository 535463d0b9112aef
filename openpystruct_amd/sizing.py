"""Batched dataset generation: the reference's `generate_sample` loop for thousands of cases at once.

Mirrors /root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py (and _MultiCore.py):
  * module-level parameters                 :20-49   -> `SizingConfig` (same names, same defaults)
  * case randomisation                      :133-160 -> `make_cases`
  * per-epoch FE solve                      :176-190 -> `beam_solve`            (HIP kernel, csrc/beam_solve.hip)
  * loss / backward / Adam / clamp / stop   :195-219 -> `ops_beam_sizing_step_f32` (HIP kernel, csrc/sizing_step.hip)
  * record assembly                         :221-249 -> `generate_dataset` / `records_to_reference_json`

All cases of a shard advance one epoch per (solve, step) pair; finished cases are frozen by the step
kernel (their solver input is no longer refreshed), so no host round trip is needed per epoch.  The host
only polls `active.any()` every `poll_every` epochs.  Multi-GPU: cases are independent, every rank owns a
contiguous slice of the globally seeded case list and nothing is exchanged (MultiCore.py:258 does the
same with processes).
"""
from __future__ import annotations

import ctypes
import os
import json
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _cabi
from .beam import BeamSolution, beam_solve


@dataclass
class SizingConfig:
    """The reference's module-level constants (SingleCore.py:20-49), same names and defaults."""
    E: float = 200e9
    nu: float = 0.3
    A: float = 0.01
    L_max: float = 200.0
    num_nodes: int = 101
    N_rollers_max: int = 4
    M_forces_max: int = 4
    L_min: float = 15
    max_force: float = -355857.0
    uniform_udl: float = -1000.0
    I_0: float = 0.5
    max_e: int = 600
    lr: float = 0.01
    gamma: float = 0.98
    alpha_moment: float = 1e-2
    alpha_shear: float = 1e-2
    tolerance: float = 5e-3
    patience: int = 5                 # SingleCore passes 5; MultiCore's default argument is 10 (MultiCore.py:130)
    random_bridge: int = 0
    roller_nodes: Tuple[int, ...] = (10, 30, 70, 85, 100)   # 1-based (SingleCore.py:62)
    zero_last_node: bool = False      # MultiCore.py:222-223 forces the last node's rotation / deflection to 0.0

    @property
    def G(self) -> float:
        return self.E / (2 * (1 + self.nu))

    @property
    def min_force(self) -> float:
        return self.max_force / 10

    @property
    def num_elements(self) -> int:
        return self.num_nodes - 1

    @classmethod
    def beam_opt(cls) -> "SizingConfig":
        """The single-case optimiser script's constants (OpenPyStruct_BeamOpt.py:23-48): UDL -5000, 1000 epochs,
        tolerance 1e-2, patience 10; its cases come from `make_beam_opt_cases`."""
        return cls(uniform_udl=-5000.0, max_e=1000, tolerance=1e-2, patience=10)

    @classmethod
    def gpu_script(cls) -> "SizingConfig":
        """OpenPyStruct_BeamOpt_training_GPU.py:50-51: tolerance 1e-2, patience 100 (otherwise as SingleCore)."""
        return cls(tolerance=1e-2, patience=100)

    def c_params(self) -> "_cabi.SizingParams":
        return _cabi.SizingParams(
            E=self.E, G=self.G, alpha_moment=self.alpha_moment, alpha_shear=self.alpha_shear, lr=self.lr,
            gamma=self.gamma, beta1=0.9, beta2=0.999, adam_eps=1e-8, clamp_min=1e-8, bend_eps=1e-6,
            area_coef=0.03, tolerance=self.tolerance, patience=self.patience, max_epochs=self.max_e)


class Cases:
    """A list of cases (what SingleCore.py:133-160 draws per sample) as tensors on one device: ragged per-case
    lists are stored zero-padded with their counts; the Python-list views the reference's records use
    (`roller_nodes`, `force_nodes`, `force_values`) are materialised lazily."""

    def __init__(self, node_positions, L, roller_nodes_t, n_rollers, force_nodes_t, n_forces, force_values_t, fix, Fy):
        self.node_positions = node_positions      # [B, N] f64 (every row equal when random_bridge == 0)
        self.L = L                                # [B] f64
        self.roller_nodes_t = roller_nodes_t      # [B, R] int64, 1-based ids, 0 = unused slot
        self.n_rollers = n_rollers                # [B] int64
        self.force_nodes_t = force_nodes_t        # [B, F] int64, 1-based ids, 0 = unused slot
        self.n_forces = n_forces                  # [B] int64
        self.force_values_t = force_values_t      # [B, F] f64, 0.0 in unused slots
        self.fix = fix                            # [B, N] uint8
        self.Fy = Fy                              # [B, N] f64
        self._lists = {}

    def __len__(self):
        return int(self.Fy.shape[0])

    def tensors(self):
        return (self.node_positions, self.L, self.roller_nodes_t, self.n_rollers, self.force_nodes_t, self.n_forces,
                self.force_values_t, self.fix, self.Fy)

    def slice(self, lo: int, hi: int) -> "Cases":
        return Cases(*(t[lo:hi] for t in self.tensors()))

    def _ragged(self, name, t, n):
        if name not in self._lists:
            rows, cnt = t.cpu().tolist(), n.cpu().tolist()
            self._lists[name] = [r[:c] for r, c in zip(rows, cnt)]
        return self._lists[name]

    @property
    def roller_nodes(self) -> List[List[int]]:
        return self._ragged("r", self.roller_nodes_t, self.n_rollers)

    @property
    def force_nodes(self) -> List[List[int]]:
        return self._ragged("f", self.force_nodes_t, self.n_forces)

    @property
    def force_values(self) -> List[List[float]]:
        return self._ragged("v", self.force_values_t, self.n_forces)


CASE_BLOCK = 16384      # the case list is drawn in blocks of this many cases, block b from the seed (seed, b)


def make_cases(n_cases: int, cfg: SizingConfig, seed: int = 20250307, device="cpu", lo: int = 0, hi: Optional[int] = None) -> Cases:
    """Cases [lo, hi) (default: all) of the globally seeded case list of `n_cases` cases.  The list is drawn in blocks of
    CASE_BLOCK cases, block b from its own generator seeded with (seed, b): case i is a pure function of (seed, i, device type)
    -- independent of n_cases, of the range asked for and of how many ranks or chunks share the list -- and a range costs
    O(hi - lo + CASE_BLOCK), not O(n_cases) (chunked generation of 10^7 cases no longer draws the whole list per chunk)."""
    hi = n_cases if hi is None else min(hi, n_cases)
    lo = max(0, min(lo, hi))
    if (torch.device(device).type == "cuda" and max(cfg.M_forces_max, cfg.N_rollers_max if cfg.random_bridge == 1 else len(cfg.roller_nodes)) <= 8
            and cfg.num_nodes >= 4):
        return _draw_cases_device(cfg, seed, torch.device(device), lo, hi)      # one launch (csrc/case_draw.hip)
    parts = []
    for b in range(lo // CASE_BLOCK, max(lo // CASE_BLOCK + 1, (hi + CASE_BLOCK - 1) // CASE_BLOCK)):
        blk = _make_case_block(cfg, seed * 1000003 + b, device)
        a, e = max(lo, b * CASE_BLOCK) - b * CASE_BLOCK, min(hi, (b + 1) * CASE_BLOCK) - b * CASE_BLOCK
        parts.append(blk.slice(a, max(a, e)))
    if len(parts) == 1:
        return parts[0]
    return Cases(*(torch.cat(ts, dim=0) for ts in zip(*(p.tensors() for p in parts))))


def _draw_cases_device(cfg: SizingConfig, seed: int, dev: torch.device, lo: int, hi: int) -> Cases:
    """Cases [lo, hi) of the list `seed` defines, drawn on the GPU by ONE launch (csrc/case_draw.hip: the same draws as
    `_make_case_block`, SingleCore.py:133-160, from a counter-based stream keyed by (seed, case number) -- the ~25 framework
    ops per block of the vectorised form are 8 % of a 50 000-case generator shard, host-bound)."""
    B, N, F = hi - lo, cfg.num_nodes, cfg.M_forces_max
    rb = int(cfg.random_bridge == 1)
    fixed = np.asarray(cfg.roller_nodes, dtype=np.int32)
    R = cfg.N_rollers_max if rb else int(fixed.size)
    f64 = dict(dtype=torch.float64, device=dev)
    i64 = dict(dtype=torch.int64, device=dev)
    Ls, nr, k = torch.empty(B, **f64), torch.empty(B, **i64), torch.empty(B, **i64)
    r_nodes, f_nodes, f_vals = torch.empty((B, R), **i64), torch.empty((B, F), **i64), torch.empty((B, F), **f64)
    fix, Fy = torch.empty((B, N), dtype=torch.uint8, device=dev), torch.empty((B, N), **f64)
    with torch.cuda.device(dev):
        rc = _cabi.load().ops_sizing_draw_cases_f64(
            B, lo, int(seed) & 0xFFFFFFFFFFFFFFFF, N, cfg.N_rollers_max, F, rb, fixed.ctypes.data, int(fixed.size),
            float(cfg.L_min), float(cfg.L_max), float(cfg.max_force), float(cfg.min_force), Ls.data_ptr(), r_nodes.data_ptr(),
            nr.data_ptr(), f_nodes.data_ptr(), k.data_ptr(), f_vals.data_ptr(), fix.data_ptr(), Fy.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream)
    if rc != _cabi.OK:
        raise RuntimeError(f"ops_sizing_draw_cases_f64 failed with code {rc}: {_cabi.load().ops_amd_last_error().decode()}")
    if rb:
        xs = torch.linspace(0.0, 1.0, N, **f64)[None, :] * Ls[:, None]
    else:
        xs = torch.linspace(0, cfg.L_max, N, **f64).expand(B, -1).contiguous()                 # SingleCore.py:59
    return Cases(xs, Ls, r_nodes, nr, f_nodes, k, f_vals, fix, Fy)


def _make_case_block(cfg: SizingConfig, seed: int, device) -> Cases:
    """One block of CASE_BLOCK cases: seeded, vectorised restatement of the case randomisation (SingleCore.py:133-160),
    generated directly on `device` (BASELINE config 3: "dataset generated on-GPU"): same distributions -- 1..4 distinct
    loaded nodes from the available ones, values U(max_force, min_force); with random_bridge = 1: L = L_min + U(0, L_max),
    1..4 distinct rollers from nodes 2..N-1.  The reference never seeds `random`."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    n_cases = CASE_BLOCK
    N, B, R, F = cfg.num_nodes, n_cases, cfg.N_rollers_max, cfg.M_forces_max
    f64 = dict(dtype=torch.float64, device=dev)
    cand = torch.arange(2, N, device=dev)                      # 1-based candidates: range(2, num_nodes), :63 / :138
    rand = lambda *shape: torch.rand(*shape, generator=g, **f64)   # noqa: E731
    if cfg.random_bridge == 1:
        Ls = cfg.L_min + rand(B) * cfg.L_max                                                   # :134
        nr = torch.randint(1, R + 1, (B,), generator=g, device=dev)                            # :139
        r_nodes = cand[torch.argsort(rand(B, cand.numel()), dim=1)[:, :R]]                     # distinct picks, :142-151
        r_used = torch.arange(R, device=dev)[None, :] < nr[:, None]
    else:
        Ls = torch.full((B,), cfg.L_max, **f64)
        r_nodes = torch.tensor(cfg.roller_nodes, device=dev).expand(B, -1)                     # :153
        r_used = torch.ones_like(r_nodes, dtype=torch.bool)
        nr = r_used.sum(dim=1)
    r_nodes = r_nodes * r_used                                  # unused slots -> 0 (a node id that does not exist)
    is_roller = torch.zeros((B, N + 1), dtype=torch.bool, device=dev).scatter_(1, r_nodes, True)
    is_roller[:, 0] = False
    # loaded nodes: distinct draws among the candidates that are not rollers (:157-159)
    score = rand(B, cand.numel()).masked_fill_(is_roller[:, 2:N], 2.0)                         # rollers sort last
    f_nodes = cand[torch.argsort(score, dim=1)[:, :F]]
    n_avail = cand.numel() - is_roller[:, 2:N].sum(dim=1)
    k = torch.minimum(torch.randint(1, F + 1, (B,), generator=g, device=dev), n_avail)        # :157-158
    f_used = torch.arange(F, device=dev)[None, :] < k[:, None]
    f_vals = (cfg.max_force + rand(B, F) * (cfg.min_force - cfg.max_force)) * f_used           # :160
    f_nodes = f_nodes * f_used
    if cfg.random_bridge == 1:
        xs = torch.linspace(0.0, 1.0, N, **f64)[None, :] * Ls[:, None]
    else:
        xs = torch.linspace(0, cfg.L_max, N, **f64).expand(B, -1).contiguous()                # SingleCore.py:59
    fix = is_roller[:, 1:].to(torch.uint8)                                                     # ops.fix(r, 0, 1, 0), :102
    fix[:, 0] = 1                                                                              # ops.fix(1, 1, 1, 0), :100
    Fy = torch.zeros((B, N + 1), **f64).scatter_add_(1, f_nodes, f_vals)[:, 1:].contiguous()   # ops.load(n, 0, F, 0), :113
    return Cases(xs, Ls, r_nodes, nr, f_nodes, k, f_vals, fix, Fy)


def make_beam_opt_cases(n_cases: int, cfg: Optional[SizingConfig] = None, seed: int = 20250307, device="cpu",
                        n_rollers: int = 5, m_forces: int = 5, min_spacing: int = 15) -> Cases:
    """Cases as OpenPyStruct_BeamOpt.py:55-80 draws its single one: `n_rollers` rollers among nodes 2..N-1 whose
    node numbers differ by at least `min_spacing` (BO:57-76 compares node ids with L_min), then `m_forces` loads
    on distinct free nodes with values U(0.5, 1) * max_force (BO:78-80).  Sequential rejection sampling per case,
    seeded; fixed 200 m geometry."""
    cfg = cfg or SizingConfig.beam_opt()
    rng = np.random.default_rng(seed)
    N = cfg.num_nodes
    cand = np.arange(2, N)
    R = np.zeros((n_cases, n_rollers), dtype=np.int64)
    Fn = np.zeros((n_cases, m_forces), dtype=np.int64)
    Fv = np.zeros((n_cases, m_forces))
    for b in range(n_cases):
        while True:      # BO:66-76 can dead-end (no node left at distance >= 15): restart the case then
            rollers = [int(rng.choice(cand))]
            ok = True
            for _ in range(1, n_rollers):
                free = [n for n in cand if all(abs(n - r) >= min_spacing for r in rollers)]
                if not free:
                    ok = False
                    break
                rollers.append(int(rng.choice(free)))
            if ok:
                break
        avail = [n for n in cand if n not in rollers]                              # BO:78
        fn = rng.choice(avail, size=min(m_forces, len(avail)), replace=False)      # BO:79
        R[b], Fn[b, : len(fn)] = rollers, fn
        Fv[b, : len(fn)] = rng.uniform(cfg.max_force, 0.5 * cfg.max_force, size=len(fn))   # BO:80
    dev = torch.device(device)
    r_nodes, f_nodes = torch.as_tensor(R, device=dev), torch.as_tensor(Fn, device=dev)
    f_vals = torch.as_tensor(Fv, device=dev)
    B = n_cases
    xs = torch.linspace(0, cfg.L_max, N, dtype=torch.float64, device=dev).expand(B, -1).contiguous()
    fix = torch.zeros((B, N + 1), dtype=torch.uint8, device=dev).scatter_(1, r_nodes, 1)[:, 1:].contiguous()
    fix[:, 0] = 1
    Fy = torch.zeros((B, N + 1), dtype=torch.float64, device=dev).scatter_add_(1, f_nodes, f_vals)[:, 1:].contiguous()
    return Cases(xs, torch.full((B,), cfg.L_max, dtype=torch.float64, device=dev), r_nodes,
                 torch.full((B,), n_rollers, dtype=torch.int64, device=dev), f_nodes, (f_nodes > 0).sum(dim=1), f_vals, fix, Fy)


def cases_from_lists(node_positions, roller_nodes, force_nodes, force_values, device="cpu") -> Cases:
    """Cases given explicitly, in the form the reference's records hold them (SingleCore.py:235-247): per case the node
    coordinates [N] (or one shared row), the 1-based roller node ids, the 1-based loaded node ids and their values.
    Supports are what `setup_model` applies: node 1 pinned, `fix(r, 0, 1, 0)` per roller (SingleCore.py:100-102)."""
    dev = torch.device(device)
    B = len(force_nodes)
    xs = torch.as_tensor(np.asarray(node_positions, dtype=np.float64), device=dev)
    if xs.dim() == 1:
        xs = xs.expand(B, -1).contiguous()
    N = int(xs.shape[1])
    if B == 0:
        raise ValueError("no cases given")
    if len(roller_nodes) == 0 or not isinstance(roller_nodes[0], (list, tuple, np.ndarray)):
        roller_nodes = [list(roller_nodes)] * B          # one support list shared by every case (possibly empty: a cantilever-free beam is singular, the solve reports it)
    if len(roller_nodes) != B or len(force_values) != B:
        raise ValueError("roller_nodes / force_nodes / force_values must have one entry per case")
    R = max(1, max(len(r) for r in roller_nodes))
    F = max(1, max(len(f) for f in force_nodes))
    rn, fn, fv = np.zeros((B, R), dtype=np.int64), np.zeros((B, F), dtype=np.int64), np.zeros((B, F))
    for b in range(B):
        if len(force_nodes[b]) != len(force_values[b]):
            raise ValueError(f"case {b}: {len(force_nodes[b])} loaded nodes but {len(force_values[b])} values")
        for name, ids in (("roller", roller_nodes[b]), ("force", force_nodes[b])):
            if any(int(n) < 1 or int(n) > N for n in ids):
                raise ValueError(f"case {b}: {name} node id outside 1..{N}")
        rn[b, :len(roller_nodes[b])] = roller_nodes[b]
        fn[b, :len(force_nodes[b])] = force_nodes[b]
        fv[b, :len(force_values[b])] = force_values[b]
    r_nodes, f_nodes, f_vals = torch.as_tensor(rn, device=dev), torch.as_tensor(fn, device=dev), torch.as_tensor(fv, device=dev)
    fix = torch.zeros((B, N + 1), dtype=torch.uint8, device=dev).scatter_(1, r_nodes, 1)[:, 1:].contiguous()
    fix[:, 0] = 1
    Fy = torch.zeros((B, N + 1), dtype=torch.float64, device=dev).scatter_add_(1, f_nodes, f_vals)[:, 1:].contiguous()
    return Cases(xs, xs[:, -1].clone(), r_nodes, (r_nodes > 0).sum(dim=1), f_nodes, (f_nodes > 0).sum(dim=1), f_vals, fix, Fy)


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous case range of `rank`: [rank*n/world, (rank+1)*n/world)."""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


_FUSED_EPOCH = os.environ.get("OPS_AMD_SIZING_FUSED", "1") == "1"     # A/B switch: 0 = separate solve and step launches
_EPOCH_TILING = int(os.environ.get("OPS_AMD_SIZING_TILING", "0"))      # lanes per beam of the sizing loop's solves (0 = rule below)


TILING_ROWS = 0x200            # include/openpystruct_amd.h OPS_AMD_TILING_ROWS: the row-staged kernels of csrc/beam_fat.hip


def sizing_tiling(n_nodes: int, shared_geometry: bool = False) -> int:
    """Lanes per beam of EVERY solve of the sizing loop (fused or not, per-epoch and final): a function of the mesh only.
    The library's own default changes kernels with the number of beams per launch; roundings -- hence early-stop epochs
    and records -- would then depend on how many ranks or chunks share the cases.  16 lanes while they fit (N <= 112):
    the row-staged kernel when geometry and constraint mask are shared by all cases (the fixed bridge), beam_solve.hip's
    otherwise (random bridges); beyond that 0 = the library's first fitting tiling, which depends on N alone."""
    if _EPOCH_TILING:
        return _EPOCH_TILING
    if n_nodes > 16 * 7:
        return 0
    return 16 | TILING_ROWS if shared_geometry else 16


class SizingState:
    """Device-resident optimiser state of a shard (what the reference keeps per sample in Python objects)."""

    def __init__(self, cases: Cases, cfg: SizingConfig, device: torch.device):
        B, N = cases.Fy.shape
        Ne = N - 1
        f64 = dict(dtype=torch.float64, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.B, self.N, self.Ne, self.cfg, self.device = B, N, Ne, cfg, device
        # one shared row when every case has the same geometry / supports (the kernel's shared-table fast path)
        shared_geom = bool((cases.node_positions == cases.node_positions[:1]).all())
        shared_fix = bool((cases.fix == cases.fix[:1]).all())
        self.x = (cases.node_positions[0] if shared_geom else cases.node_positions).to(**f64).contiguous()
        self.fix = (cases.fix[0] if shared_fix else cases.fix).to(dtype=torch.uint8, device=device).contiguous()
        self.Fy = cases.Fy.to(**f64).contiguous()
        self.E = torch.tensor(cfg.E, **f64)
        self.wy = torch.tensor(cfg.uniform_udl, **f64)
        self.I = torch.full((B, Ne), cfg.I_0, **f32)                    # I_tensor, :163
        # fused epochs keep no widened copy: the kernel records, once per case, the float32 inertias of its last solve
        self._fused = _FUSED_EPOCH and Ne <= 128
        self.I_last = torch.full((B, Ne), cfg.I_0, **f32) if self._fused else None
        self.I64 = None if self._fused else self.I.double()      # separate solve + step launches: the solver's input
        self.exp_avg = torch.zeros((B, Ne), **f32)
        self.exp_avg_sq = torch.zeros((B, Ne), **f32)
        self.best_loss = torch.full((B,), float("inf"), **f32)          # :170
        self.patience_cnt = torch.zeros((B,), dtype=torch.int32, device=device)
        self.epochs_run = torch.zeros((B,), dtype=torch.int32, device=device)
        self.active = torch.ones((B,), dtype=torch.uint8, device=device)
        self.last_loss = torch.zeros((B,), **f32)
        self.V32: Optional[torch.Tensor] = None      # set by finalize()
        self.M32: Optional[torch.Tensor] = None
        self.sol: Optional[BeamSolution] = None
        self._V = torch.empty((B, Ne), **f32)         # element end forces of the epoch's solve, rounded like :189-190
        self._M = torch.empty((B, Ne), **f32)
        self._status = torch.zeros((B,), dtype=torch.int32, device=device)
        self._hp = cfg.c_params()
        sched = np.zeros((max(int(cfg.max_e), 1), 2), dtype=np.float32)          # per-epoch step size / bias correction
        _cabi.load().ops_sizing_schedule_f32(ctypes.byref(self._hp), sched.ctypes.data)
        self._schedule = torch.as_tensor(sched, device=device)

    def reset(self, cases: Cases, cfg: SizingConfig) -> bool:
        """Re-arm this state IN PLACE for another shard of the same shape (the buffers a captured epoch graph points at stay
        where they are).  False: the new cases do not fit these buffers (shape, shared geometry / supports, hyper-parameters)."""
        if tuple(cases.Fy.shape) != (self.B, self.N) or bytes(cfg.c_params()) != bytes(self._hp) or int(cfg.max_e) != int(self.cfg.max_e):
            return False
        shared_geom = bool((cases.node_positions == cases.node_positions[:1]).all())
        shared_fix = bool((cases.fix == cases.fix[:1]).all())
        if shared_geom != (self.x.dim() == 1) or shared_fix != (self.fix.dim() == 1):
            return False
        self.cfg = cfg
        # E and the line load are device scalars the captured launches read: c_params() carries neither the load nor (exactly) E
        self.E.fill_(cfg.E)
        self.wy.fill_(cfg.uniform_udl)
        self.x.copy_(cases.node_positions[0] if shared_geom else cases.node_positions)
        self.fix.copy_(cases.fix[0] if shared_fix else cases.fix)
        self.Fy.copy_(cases.Fy)
        self.I.fill_(cfg.I_0)
        if self.I_last is not None:
            self.I_last.fill_(cfg.I_0)
        else:
            self.I64.fill_(float(np.float32(cfg.I_0)))
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        self.best_loss.fill_(float("inf"))
        self.patience_cnt.zero_(); self.epochs_run.zero_(); self.active.fill_(1); self.last_loss.zero_(); self._status.zero_()
        self.V32 = self.M32 = None
        return True

    def epoch(self) -> None:
        """One epoch for every case of the shard: FE solve (:176-190) then optimiser step (:195-219).  Inside the loop
        the reference reads only eleResponse: the solve writes forces only and skips wavefronts of finished cases."""
        lib = _cabi.load()
        N, Ne = self.N, self.Ne
        if self._fused:
            # solve + optimiser step in ONE launch: the wavefront that solved a case steps it on the forces it still holds in LDS
            with torch.cuda.device(self.device):
                rc = lib.ops_beam_sizing_epoch_f32(
                    self.B, Ne, self.x.data_ptr(), N if self.x.dim() == 2 else 0, self.E.data_ptr(), 0,
                    self.fix.data_ptr(), N if self.fix.dim() == 2 else 0, self.Fy.data_ptr(), N, self.wy.data_ptr(), 0,
                    self.I.data_ptr(), self.I_last.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                    self.best_loss.data_ptr(), self.patience_cnt.data_ptr(), self.epochs_run.data_ptr(), self.active.data_ptr(),
                    self.last_loss.data_ptr(), ctypes.byref(self._hp), self._schedule.data_ptr(), self._status.data_ptr(), self._tiling(),
                    torch.cuda.current_stream(self.device).cuda_stream)
            if rc != _cabi.OK:
                raise RuntimeError(f"ops_beam_sizing_epoch_f32 failed with code {rc}: {lib.ops_amd_last_error().decode()}")
            return
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = lib.ops_beam_solve_forces_f32(
                self.B, Ne, self.x.data_ptr(), N if self.x.dim() == 2 else 0, self.E.data_ptr(), 0, self.I64.data_ptr(), Ne,
                self.fix.data_ptr(), N if self.fix.dim() == 2 else 0, self.Fy.data_ptr(), N, self.wy.data_ptr(), 0,
                self._V.data_ptr(), self._M.data_ptr(), self._status.data_ptr(), self.active.data_ptr(), sizing_tiling(N), stream)   # float32 forces: beam_solve.hip only
            if rc != _cabi.OK:
                raise RuntimeError(f"ops_beam_solve_forces_f32 failed with code {rc}: {lib.ops_amd_last_error().decode()}")
            rc = lib.ops_beam_sizing_step_vm32_f32(
                self.B, Ne, self.I.data_ptr(), self.I64.data_ptr(), self._V.data_ptr(), self._M.data_ptr(),
                self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.best_loss.data_ptr(),
                self.patience_cnt.data_ptr(), self.epochs_run.data_ptr(), self.active.data_ptr(),
                self.last_loss.data_ptr(), ctypes.byref(self._hp), self._schedule.data_ptr(), stream)
        if rc != _cabi.OK:
            raise RuntimeError(f"ops_beam_sizing_step_vm32_f32 failed with code {rc}")

    def _tiling(self) -> int:
        return sizing_tiling(self.N, shared_geometry=self.x.dim() == 1 and self.fix.dim() == 1)

    def finalize(self) -> None:
        """What the reference reads after the loop (:224-232) and records (:239-249): the state of every case's LAST
        solve.  `I64` (separate launches) froze when a case stopped / `I_last` (fused epochs) recorded its inertias, so one full solve reproduces it -- displacements included -- and the
        float32 roundings of shear / moment (:189-190) are taken from it."""
        # same tiling as the epochs, whatever the shard size: the records do not depend on how many GPUs share the cases
        if self._fused:
            self.I64 = self.I_last.double()
        self.sol = beam_solve(self.x, self.E, self.I64, self.fix, self.Fy, self.wy, tiling=self._tiling() if self._fused else sizing_tiling(self.N),
                              out=self.sol)
        self.V32, self.M32 = self.sol.V.float(), self.sol.M.float()


# The most recent (state, captured epoch graph) per device and thread: a generator run is many shards of one shape
# (`generate_dataset_to_files` chunks, the ranks' equal shards), and capture + instantiation of the 25-epoch graph costs
# ~2.2 ms of a 27 ms shard of 50 000 cases (scripts/generator_breakdown.py).  One entry per key bounds the memory held.
_EPOCH_GRAPHS: Dict[tuple, tuple] = {}
_POLL_FLAGS: Dict[tuple, tuple] = {}


def optimize_cases(cases: Cases, cfg: SizingConfig, device, poll_every: int = 25, use_graph: bool = True,
                   reuse: bool = False, record_loss: bool = False) -> SizingState:
    """Run the sizing loop of every case to its early stop (or max_e).  Returns the final device state.  With `reuse` the
    state buffers and the captured graph of the previous shard of the same shape are re-armed in place: tensors of an
    earlier returned state are then overwritten (`generate_dataset` copies what it hands out).  `record_loss` keeps every
    epoch's `total_loss` (SingleCore.py:199) per case in `state.loss_history` [epochs, B] (rows past a case's `epochs_run`
    repeat its last value); it runs the loop launch by launch."""
    device = torch.device(device)
    if record_loss:
        use_graph, reuse = False, False
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device() if device.type == "cuda" else -1,
           threading.get_ident(), int(poll_every))
    st, graph = (None, None)
    if reuse and use_graph and poll_every > 1 and key in _EPOCH_GRAPHS:
        st, graph = _EPOCH_GRAPHS[key]
        if not st.reset(cases, cfg):
            st, graph = None, None
            del _EPOCH_GRAPHS[key]
    if st is None:
        st = SizingState(cases, cfg, device)
    if st.B == 0:
        return st
    epochs_done = 0
    if graph is None and use_graph and poll_every > 1:
        # the epoch body is launch-bound (two short kernels): replay `poll_every` epochs as one HIP graph
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            st.epoch()                      # warm-up outside capture (allocates the solution buffers)
            epochs_done = 1
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                for _ in range(poll_every):
                    st.epoch()
        torch.cuda.current_stream(device).wait_stream(side)
        if reuse:
            _EPOCH_GRAPHS[key] = (st, graph)
    if graph is not None:
        # The "any case still active?" poll lags one replay: its answer travels through a pinned flag behind an event, and the
        # host reads the flag of replay k - 1 after it has queued replay k -- the device never waits for the host (a blocking
        # poll drained the queue 13 times per 50 000-case shard, ~40 us each); the price is one replay of skipped wavefronts.
        if key not in _POLL_FLAGS:          # (pinned allocations are slow: one pair of flags and events per device and thread)
            _POLL_FLAGS[key] = (torch.zeros(2, dtype=torch.uint8).pin_memory(), [torch.cuda.Event(), torch.cuda.Event()])
        flags, events = _POLL_FLAGS[key]
        flags.zero_()
        k = 0
        while epochs_done < cfg.max_e:
            graph.replay()
            epochs_done += poll_every
            flags[k & 1: (k & 1) + 1].copy_(st.active.any().to(torch.uint8).reshape(1), non_blocking=True)
            events[k & 1].record()
            lag = 0 if os.environ.get("OPS_AMD_SIZING_POLL_LAG", "1") == "0" else 1
            if k >= lag:
                events[(k - lag) & 1].synchronize()
                if int(flags[(k - lag) & 1]) == 0:
                    break
            k += 1
    else:
        hist = []
        while epochs_done < cfg.max_e:
            st.epoch()
            epochs_done += 1
            if record_loss:
                hist.append(st.last_loss.clone())
            if epochs_done % poll_every:
                continue
            if not bool(st.active.any()):       # host sync once per `poll_every` epochs
                break
        if record_loss:
            st.loss_history = torch.stack(hist) if hist else torch.zeros((0, st.B), dtype=torch.float32, device=device)
    # a case that is still active here ran out of max_e inside the step kernel already (it clears `active`)
    st.finalize()
    torch.cuda.synchronize(device)
    return st


RECORD_KEYS = ("roller_x_locations", "force_x_locations", "force_values", "I_values", "shear_forces",
               "bending_moments", "node_positions", "roller_nodes", "force_nodes", "num_nodes", "L",
               "rotations", "deflections")   # SingleCore.py:73-87


def generate_dataset(n_cases: int, cfg: Optional[SizingConfig] = None, device="cuda", seed: int = 20250307,
                     rank: int = 0, world: int = 1, poll_every: int = 25,
                     case_range: Optional[Tuple[int, int]] = None) -> Dict[str, object]:
    """The reference's `main()` (SingleCore.py:251-264) for this rank's shard: returns the 13 record fields
    (SingleCore.py:235-249) as tensors / lists, plus `epochs_run`, `status` and the global case ids."""
    cfg = cfg or SizingConfig()
    lo, hi = shard_range(n_cases, rank, world)
    if case_range is not None:               # a sub-range of this rank's shard (generate_dataset_to_files)
        lo, hi = lo + case_range[0], min(hi, lo + case_range[1])
    cases = make_cases(n_cases, cfg, seed, device=device, lo=lo, hi=hi)     # generated on the GPU: only the blocks this range touches
    st = optimize_cases(cases, cfg, device, poll_every=poll_every, reuse=True)     # everything handed out below is a copy
    sol = st.sol
    rot, defl = sol.theta.clone(), sol.v.clone()
    if cfg.zero_last_node:
        rot[:, -1] = 0.0
        defl[:, -1] = 0.0
    xs = cases.node_positions
    x_at = lambda nodes: torch.gather(xs, 1, (nodes - 1).clamp_min(0)) * (nodes > 0)   # noqa: E731  zero-padded like the ids
    return {
        # ragged fields as zero-padded tensors + counts: exactly the arrays the reference's pad_sequences builds
        # (PINN:66-76); `records_to_reference_json` turns them back into the ragged lists of the wire format
        "roller_x_locations": x_at(cases.roller_nodes_t),
        "force_x_locations": x_at(cases.force_nodes_t),
        "force_values": cases.force_values_t,
        "I_values": st.I.clone(),                 # float32, AFTER the last Adam step (:239); (copies: the state is re-armed in place by the next shard)
        "shear_forces": st.V32,                   # float32, state of the last solve (:240)
        "bending_moments": st.M32,
        "node_positions": xs,
        "roller_nodes": cases.roller_nodes_t,
        "force_nodes": cases.force_nodes_t,
        "num_nodes": cfg.num_nodes,
        "L": cases.L,
        "rotations": rot,
        "deflections": defl,
        "n_rollers": cases.n_rollers,
        "n_forces": cases.n_forces,
        "epochs_run": st.epochs_run.clone(),
        "status": sol.status.clone(),
        # not a reference field: the float64 inertias the recorded V / M / rotations / deflections were solved with
        # (the state BEFORE the last Adam step; `I_values` is the state after it -- the reference's one-step lag)
        "I_solved": st.I64 if st._fused else st.I64.clone(),
        # (numpy: a framework arange over more than 32 768 elements wakes the whole CPU thread pool -- inside a container with a CPU
        #  quota that froze the process for the rest of the scheduler period, once per shard: runtime.py item 1)
        "case_ids": torch.from_numpy(np.arange(lo, hi, dtype=np.int64)),
    }


def records_to_reference_json(rec: Dict[str, object], path: str, drop_failed: bool = True) -> int:
    """Writes the reference's wire format (SingleCore.py:73-87, :263-264): a JSON object of 13 parallel
    lists.  `drop_failed` mirrors MultiCore.py:265 (samples whose analysis failed are filtered out).
    Ragged fields may be Python lists or zero-padded tensors with `n_rollers` / `n_forces` counts."""
    B = int(rec["I_values"].shape[0])
    status = rec["status"].cpu().numpy() if torch.is_tensor(rec.get("status")) else np.zeros(B, dtype=np.int32)
    keep = [b for b in range(B) if not (drop_failed and status[b] != 0)]

    def rows(t):
        a = t.detach().cpu().numpy()
        return [a[b].tolist() for b in keep]

    def ragged(key, count_key):
        v = rec[key]
        if not torch.is_tensor(v):
            return [v[b] for b in keep]
        a, n = v.detach().cpu().tolist(), rec[count_key].cpu().tolist()
        return [a[b][: n[b]] for b in keep]

    Lall = rec["L"].cpu().tolist() if torch.is_tensor(rec["L"]) else list(rec["L"])
    out = {
        "roller_x_locations": ragged("roller_x_locations", "n_rollers"),
        "force_x_locations": ragged("force_x_locations", "n_forces"),
        "force_values": ragged("force_values", "n_forces"),
        "I_values": rows(rec["I_values"]),
        "shear_forces": rows(rec["shear_forces"]),
        "bending_moments": rows(rec["bending_moments"]),
        "node_positions": rows(rec["node_positions"]),
        "roller_nodes": ragged("roller_nodes", "n_rollers"),
        "force_nodes": ragged("force_nodes", "n_forces"),
        "num_nodes": [int(rec["num_nodes"])] * len(keep),
        "L": [float(Lall[b]) for b in keep],
        "rotations": rows(rec["rotations"]),
        "deflections": rows(rec["deflections"]),
    }
    with open(path, "w") as f:
        json.dump(out, f)
    return len(keep)


def records_from_reference_json(path: str) -> Dict[str, object]:
    """Reads the reference's wire format back (the `json.load` at PINN:192-196 / TFD:240-244): the 13 parallel lists,
    with the fixed-width fields as tensors and the ragged ones zero-padded + counted, i.e. the shape
    `generate_dataset` returns and `dataprep.prepare` consumes.  Raises KeyError on a missing field."""
    with open(path) as f:
        d = json.load(f)
    missing = [k for k in RECORD_KEYS if k not in d]
    if missing:
        raise KeyError(f"{path}: missing record fields {missing}")
    B = len(d["I_values"])

    def padded(key, dtype):
        w = max((len(r) for r in d[key]), default=0)
        t = torch.zeros((B, w), dtype=dtype)
        for b, r in enumerate(d[key]):
            t[b, : len(r)] = torch.as_tensor(r, dtype=dtype)
        return t

    rec: Dict[str, object] = {
        "roller_x_locations": padded("roller_x_locations", torch.float64),
        "force_x_locations": padded("force_x_locations", torch.float64),
        "force_values": padded("force_values", torch.float64),
        "roller_nodes": padded("roller_nodes", torch.int64),
        "force_nodes": padded("force_nodes", torch.int64),
        "n_rollers": torch.tensor([len(r) for r in d["roller_nodes"]], dtype=torch.int64),
        "n_forces": torch.tensor([len(r) for r in d["force_nodes"]], dtype=torch.int64),
        "I_values": torch.tensor(d["I_values"], dtype=torch.float32),
        "shear_forces": torch.tensor(d["shear_forces"], dtype=torch.float32),
        "bending_moments": torch.tensor(d["bending_moments"], dtype=torch.float32),
        "node_positions": torch.tensor(d["node_positions"], dtype=torch.float64),
        "rotations": torch.tensor(d["rotations"], dtype=torch.float64),
        "deflections": torch.tensor(d["deflections"], dtype=torch.float64),
        "L": torch.tensor(d["L"], dtype=torch.float64),
        "num_nodes": int(d["num_nodes"][0]) if B else 0,
        "status": torch.zeros(B, dtype=torch.int32),
    }
    return rec


def save_records(rec: Dict[str, object], path: str) -> None:
    """Binary fast path for datasets that stay inside this package: one `torch.save` of the record
    tensors moved to the host -- 200 000 cases are ~0.5 GB here against several GB of JSON text."""
    torch.save({k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in rec.items()}, path)


def load_records(path: str, device="cpu") -> Dict[str, object]:
    rec = torch.load(path, map_location=device, weights_only=True)
    missing = [k for k in RECORD_KEYS if k not in rec]
    if missing:
        raise KeyError(f"{path}: missing record fields {missing}")
    return rec


def generate_dataset_to_files(n_cases: int, out_dir: str, cfg: Optional[SizingConfig] = None, device="cuda", seed: int = 20250307,
                              rank: int = 0, world: int = 1, chunk: int = 100000, resume: bool = True) -> List[str]:
    """`generate_dataset` for this rank's shard, flushed chunk by chunk (`save_records`) so that an interrupted run keeps
    what it has -- the reference writes ONE json at the very end (SingleCore.py:263) and loses everything on a crash
    (SURVEY section 5).  With `resume`, chunks whose file exists are skipped after checking that the file records the same
    (seed, n_cases, configuration); case i is a pure function of (seed, i), so the files of any run / any GPU count tile the
    same dataset.  Returns the chunk files in order."""
    os.makedirs(out_dir, exist_ok=True)
    lo, hi = shard_range(n_cases, rank, world)
    files = []
    for c0 in range(0, hi - lo, chunk):
        c1 = min(c0 + chunk, hi - lo)
        path = os.path.join(out_dir, f"records_{lo + c0:09d}_{lo + c1:09d}.pt")
        files.append(path)
        meta = {"seed": int(seed), "n_cases": int(n_cases), "cfg": repr(cfg or SizingConfig()), "range": [lo + c0, lo + c1]}
        if resume and os.path.exists(path):
            have = torch.load(path, map_location="cpu", weights_only=True).get("meta")
            if have != meta:      # a file of another dataset (seed / size / configuration): never mix silently
                raise ValueError(f"{path} was generated with {have}, this run is {meta}; use another out_dir or resume=False")
            continue
        rec = generate_dataset(n_cases, cfg, device, seed, rank, world, case_range=(c0, c1))
        rec["meta"] = meta
        save_records(rec, path + ".tmp")
        os.replace(path + ".tmp", path)          # a file either is complete or does not exist
    return files


def concat_records(records: List[Dict[str, object]]) -> Dict[str, object]:
    """Chunks (as loaded by `load_records`) -> one record set in the tensor form `dataprep.prepare` consumes."""
    out: Dict[str, object] = {}
    for k, v in records[0].items():
        if torch.is_tensor(v) and v.dim() > 0:
            width = max(r[k].shape[1] for r in records) if v.dim() == 2 else None
            parts = [torch.nn.functional.pad(r[k], (0, width - r[k].shape[1])) if width is not None and r[k].shape[1] < width else r[k]
                     for r in records]
            out[k] = torch.cat(parts, dim=0)
        else:
            out[k] = v
    return out
