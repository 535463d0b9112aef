import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from openpystruct_amd import _cabi, frames
sys.path.insert(0, 'scripts')
from frame_coop_sweep import timed
lib = _cabi.load()
for (bays, stories, B) in [(10,10,1),(10,10,256),(10,10,512),(10,10,768),(10,10,1024),(12,12,512),(13,14,256),(15,16,1),(15,16,256),(15,16,512),(15,16,768),(9,9,512),(5,5,512)]:
    topo = frames.grid_frame(bays, stories)
    I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda") * (1 + 0.1 * torch.rand(B, topo.Ne, dtype=torch.float64, device="cuda"))
    out = {}
    for name, coop in (("r06_default", 1), ("without_coop", 0)):
        _cabi.set_option("frame_coop", coop)
        topo.__dict__.pop("_ws", None)
        sol = frames.frame_solve(topo, I); torch.cuda.synchronize()
        out[name] = (int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd)) >> 24, round(timed(topo, I, sol), 1))
    _cabi.set_option("frame_coop", 1)
    print(f"{bays}x{stories} B={B}: default family {out['r06_default'][0]} {out['r06_default'][1]} us | without the four-wave kernel: family {out['without_coop'][0]} {out['without_coop'][1]} us", flush=True)
