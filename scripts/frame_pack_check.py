#!/usr/bin/env python3
"""Packed frame kernel (csrc/frame_pack.hpp: 16 / 32 lanes per frame) against the oracle on the shapes of the reference's random range
(FR:17-18: bays, stories ~ U{1..10}), and its time per launch against one wave per frame (library option frame_pack = 0: the 36-wide register window of csrc/frame_wave.hpp;
r05's 16- and 24-wide instantiations of that kernel are gone, their numbers are in profiles/r06_pack_check_first.log).

    python scripts/frame_pack_check.py            # parity on 20 shapes + A/B on 6
    python scripts/frame_pack_check.py ab          # A/B only
    python scripts/frame_pack_check.py ab 4x4:32768 9x4:16384    # A/B on these (bays x stories : frames per launch)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import _cabi, frames  # noqa: E402

_cabi.set_option("frame_latency_batch", 0)


def check():
    from oracle import beam_oracle as bo
    shapes = [(1, 1), (1, 2), (2, 1), (1, 10), (10, 1), (2, 2), (2, 3), (3, 2), (3, 3), (10, 2), (3, 10), (4, 4), (9, 4), (5, 5), (10, 5), (6, 6),
              (7, 7), (7, 9), (10, 7), (8, 8), (8, 10), (10, 8), (9, 9), (10, 9), (10, 10), (12, 12), (13, 14), (15, 16), (20, 17), (18, 20)]
    worst = 0.0
    for bays, stories in shapes:
        topo = frames.grid_frame(bays, stories)
        rng = np.random.default_rng(bays * 100 + stories)
        B = 11
        I = np.exp(rng.uniform(np.log(5e-5), np.log(5e-3), size=(B, topo.Ne)))
        I[6, min(3, topo.Ne - 1)] = -1.0          # one frame of the second wave / group is not positive definite
        loads = np.broadcast_to(topo.nodal_loads, (B,) + topo.nodal_loads.shape).copy()
        loads[9, -1, 0] = np.nan                 # one frame has a NaN load
        sol = frames.frame_solve(topo, torch.as_tensor(I, device="cuda"), torch.as_tensor(loads, device="cuda"))
        torch.cuda.synchronize()
        st = sol.status.cpu().numpy()
        ok = st[6] != 0 and st[[i for i in range(B) if i != 6]].sum() == 0
        errs = []
        for b in (0, 1, 4, 5, 7, 8, 10):
            d, f, s_, neq, kd = bo.solve_model_3dof(topo.coords, topo.conn, topo.A, topo.E, I[b], topo.fix3, topo.nodal_loads, wy=topo.wy, wx=topo.wx)
            ed = np.abs(sol.disp[b].cpu().numpy() - d).max() / np.abs(d).max()
            ef = np.abs(sol.forces[b].cpu().numpy() - f).max() / np.abs(f).max()
            errs.append(max(ed, ef / 10))
        e = max(errs)
        worst = max(worst, e)
        nan9 = bool(torch.isnan(sol.disp[9]).any())
        print(f"{bays:2d} x {stories:2d}  n_eq {topo.n_eq:4d}  kd {topo.kd:2d}  max rel err {e:.2e}  status ok {ok}  nan frame isolated {nan9 and bool(torch.isfinite(sol.disp[8]).all()) and bool(torch.isfinite(sol.disp[10]).all())}",
              flush=True)
        assert ok and e < 1e-7, (bays, stories, st, errs)
    print("worst", worst)


def ab(cases=None):
    cases = cases or [(1, 10, 131072), (2, 2, 131072), (3, 3, 65536), (10, 3, 32768), (5, 5, 32768), (7, 7, 16384), (8, 8, 16384), (9, 9, 16384), (8, 10, 16384), (10, 10, 16384), (12, 12, 12288), (15, 16, 12288)]
    for bays, stories, B in cases:
        topo = frames.grid_frame(bays, stories)
        I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda") * (1 + 0.1 * torch.rand(B, topo.Ne, dtype=torch.float64, device="cuda"))
        rec = {"frame": f"{bays}x{stories}", "n_eq": topo.n_eq, "half_bandwidth": topo.kd, "B": B}
        outs = {}
        for name, val in (("wave", "0"), ("pack", "1")):
            _cabi.set_option("frame_pack", int(val))
            topo.__dict__.pop("_ws", None)
            sol = frames.frame_solve(topo, I)
            torch.cuda.synchronize()
            assert int(sol.status.abs().sum()) == 0
            outs[name] = sol.disp.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(5):
                    frames.frame_solve(topo, I, out=sol)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            rec[name + "_ms"] = round(best, 4)
            rec[name + "_solves_per_s"] = round(B / best * 1e3)
        rec["speedup"] = round(rec["wave_ms"] / rec["pack_ms"], 3)
        rec["max_rel_diff"] = float((outs["wave"] - outs["pack"]).abs().max() / outs["wave"].abs().max())
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] != "ab":
        check()
    # python scripts/frame_pack_check.py ab 4x4:32768 9x4:16384      (A/B on the named shapes only; OPS_AMD_LIB=<variant> selects another build)
    ab([(int(a.split("x")[0]), int(a.split("x")[1].split(":")[0]), int(a.split(":")[1])) for a in sys.argv[2:]] or None)
