#!/usr/bin/env python3
"""What the drop-in costs when it is used the way the reference's scripts use OpenSees: ONE model per call.
Beam (100 elements) and 10 x 10 frame: time per solve at batch 1 through the library, per `analyze` through the command shim
(model rebuilt by Python commands every time, as setup_model / setup_frame_model do), per sizing epoch at batch 1, beside the CPU
oracle's solve and the batched rate.  -> gpurun_out/batch_of_one_latency.json"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openpystruct_amd as oa
from openpystruct_amd import frames, ops, sizing
from oracle import beam_oracle as bo      # (a script under scripts/: the checker's CPU time beside the product's, never the product)

dev = torch.device("cuda:0")
res = {}

def timed(fn, n, sync=True):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# ---- beam, 100 elements ----
cfg = sizing.SizingConfig()
x = torch.linspace(0, cfg.L_max, cfg.num_nodes, dtype=torch.float64, device=dev)
I1 = torch.full((1, cfg.num_nodes - 1), 0.5, dtype=torch.float64, device=dev)
Fy = torch.zeros((1, cfg.num_nodes), dtype=torch.float64, device=dev); Fy[0, 50] = -1e5
fixh = bo.reference_fix_mask(cfg.num_nodes, tuple(cfg.roller_nodes))
fix = torch.as_tensor(np.asarray(fixh, dtype=np.uint8), device=dev)[None]
E, wy = float(cfg.E), float(cfg.uniform_udl)
out = oa.beam_solve(x=x, E=E, I=I1, fix=fix[0], Fy=Fy, wy=wy)
res["beam_solve_B1_with_sync_us"] = 1e6 * timed(lambda: (oa.beam_solve(x=x, E=E, I=I1, fix=fix[0], Fy=Fy, wy=wy, out=out), torch.cuda.synchronize()), 200)
res["beam_solve_B1_queued_us"] = 1e6 * timed(lambda: oa.beam_solve(x=x, E=E, I=I1, fix=fix[0], Fy=Fy, wy=wy, out=out), 2000)
xs = np.linspace(0, cfg.L_max, cfg.num_nodes)
def shim_beam():
    ops.wipe(); ops.model('basic', '-ndm', 2, '-ndf', 3)
    for i, xi in enumerate(xs):
        ops.node(i + 1, float(xi), 0.0)
    ops.fix(1, 1, 1, 0)
    for r in cfg.roller_nodes:
        ops.fix(int(r), 0, 1, 0)
    ops.geomTransf('Linear', 1)
    for e in range(cfg.num_nodes - 1):
        ops.element('elasticBeamColumn', e + 1, e + 1, e + 2, cfg.A, cfg.E, 0.5, 1)
    ops.timeSeries('Linear', 1); ops.pattern('Plain', 1, 1); ops.load(51, 0.0, -1e5, 0.0)
    for e in range(cfg.num_nodes - 1):
        ops.eleLoad('-ele', e + 1, '-type', '-beamUniform', cfg.uniform_udl)
    ops.system('BandSPD'); ops.numberer('RCM'); ops.constraints('Plain'); ops.integrator('LoadControl', 1.0)
    ops.algorithm('Linear'); ops.analysis('Static')
    assert ops.analyze(1) == 0
    return [ops.eleResponse(e + 1, 'forces')[2] for e in range(cfg.num_nodes - 1)]
ops.set_device(dev)
res["shim_beam_rebuild_analyze_read_us"] = 1e6 * timed(shim_beam, 50, sync=False)
def shim_beam_analyze_only():
    assert ops.analyze(1) == 0
shim_beam()
res["shim_beam_analyze_only_us"] = 1e6 * timed(shim_beam_analyze_only, 200, sync=False)
Fyh = Fy[0].cpu().numpy()
t0 = time.perf_counter()
for _ in range(200):
    bo.solve_beam_dense(xs, cfg.E, np.full(cfg.num_nodes - 1, 0.5), np.asarray(fixh, dtype=np.uint8), Fyh, cfg.uniform_udl)
res["cpu_oracle_beam_dense_us"] = 1e6 * (time.perf_counter() - t0) / 200
st = sizing.optimize_cases(sizing.make_cases(1, cfg, device=dev), cfg, dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
st = sizing.optimize_cases(sizing.make_cases(1, cfg, device=dev), cfg, dev)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
res["sizing_B1_epochs"] = int(st.epochs_run.max()); res["sizing_B1_us_per_epoch"] = 1e6 * dt / max(1, int(st.epochs_run.max()))

# ---- frame, 10 x 10 (the script's largest) ----
topo = frames.grid_frame(10, 10, device=dev)
If = torch.full((1, topo.Ne), 5e-4, dtype=torch.float64, device=dev)
sol = frames.frame_solve(topo, If)
res["frame_solve_B1_with_sync_us"] = 1e6 * timed(lambda: (frames.frame_solve(topo, If, out=sol), torch.cuda.synchronize()), 100)
res["frame_solve_B1_queued_us"] = 1e6 * timed(lambda: frames.frame_solve(topo, If, out=sol), 500)
t0 = time.perf_counter()
for _ in range(20):
    bo.solve_model_3dof(topo.coords, topo.conn, topo.A, topo.E, np.full(topo.Ne, 5e-4), topo.fix3, topo.nodal_loads, wy=topo.wy, wx=topo.wx)
res["cpu_oracle_frame_us"] = 1e6 * (time.perf_counter() - t0) / 20
def shim_frame():
    ops.wipe(); ops.model('basic', '-ndm', 2, '-ndf', 3)
    for i, (cx, cy) in enumerate(topo.coords):
        ops.node(i + 1, float(cx), float(cy))
    for i, f in enumerate(topo.fix3):
        if f.any():
            ops.fix(i + 1, int(f[0]), int(f[1]), int(f[2]))
    ops.geomTransf('Linear', 1)
    for e, (a_, b_) in enumerate(topo.conn):
        ops.element('elasticBeamColumn', e + 1, int(a_) + 1, int(b_) + 1, float(topo.A[e]), float(topo.E[e]), 5e-4, 1)
    ops.timeSeries('Linear', 1); ops.pattern('Plain', 1, 1)
    for i, l in enumerate(topo.nodal_loads):
        if np.any(l != 0):
            ops.load(i + 1, float(l[0]), float(l[1]), float(l[2]))
    wy_, wx_ = np.broadcast_to(topo.wy, (topo.Ne,)), np.broadcast_to(topo.wx, (topo.Ne,))
    for e in range(topo.Ne):
        if wy_[e] != 0 or wx_[e] != 0:
            ops.eleLoad('-ele', e + 1, '-type', '-beamUniform', float(wy_[e]), float(wx_[e]))
    ops.system('BandGeneral'); ops.numberer('RCM'); ops.constraints('Plain'); ops.integrator('LoadControl', 1.0)
    ops.algorithm('Newton'); ops.analysis('Static')
    assert ops.analyze(1) == 0
    return [ops.eleResponse(e + 1, 'forces')[2] for e in range(topo.Ne)]
res["shim_frame_rebuild_analyze_read_us"] = 1e6 * timed(shim_frame, 30, sync=False)
shim_frame()
res["shim_frame_analyze_only_us"] = 1e6 * timed(shim_beam_analyze_only, 100, sync=False)
_ = frames.optimize_frames(topo, 1, max_epochs=50)
torch.cuda.synchronize(); t0 = time.perf_counter()
I_, s_, ep = frames.optimize_frames(topo, 1, max_epochs=300)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
res["frame_sizing_B1_epochs"] = int(ep.max()); res["frame_sizing_B1_us_per_epoch"] = 1e6 * dt / max(1, int(ep.max()))
for B in (1024, 16384):
    IfB = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device=dev)
    solB = frames.frame_solve(topo, IfB)
    res[f"frame_solve_B{B}_us_per_frame"] = 1e6 * timed(lambda: frames.frame_solve(topo, IfB, out=solB), 20) / B
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/batch_of_one_latency.json", "w"), indent=1)
