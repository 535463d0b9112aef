#!/usr/bin/env python3
"""Lane-per-beam alternative vs the product kernel (same inputs, HIP-event timing of graphs of launches)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import openpystruct_amd as oa
from openpystruct_amd import _cabi
lib = _cabi.load()
for B, K in ((10000, 200), (1 << 20, 10)):
    inp = bench.synth_inputs(B, 0, torch.device("cuda"), "trajectory")
    out = oa.beam_solve(**inp)
    nbytes = int(lib.ops_beam_solve_lane_workspace_bytes(B, 100))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = out.status
    def lane():
        rc = lib.ops_beam_solve_lane_per_beam_f64(B, 100, inp["x"].data_ptr(), inp["E"].data_ptr(), inp["I"].data_ptr(), inp["fix"].data_ptr(),
                                                  inp["Fy"].data_ptr(), inp["wy"].data_ptr(), out.v.data_ptr(), out.theta.data_ptr(),
                                                  out.V.data_ptr(), out.M.data_ptr(), st.data_ptr(), ws.data_ptr(), nbytes,
                                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    def prod():
        oa.beam_solve(**inp, out=out)
    res = {"beams": B}
    for name, fn in (("product", prod), ("lane_per_beam", lane)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): fn()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(K): fn()
            g.replay(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); s.synchronize()
        res[name + "_us"] = e0.elapsed_time(e1) * 1e3 / K
    print(json.dumps(res))
