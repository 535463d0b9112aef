#!/bin/bash
# A/B of variants of the wave-per-frame kernel (ab/lib_fw_<name>.so, scripts/build_frame_variant.sh) against the product library: ms per launch
cd "$GRAFT_REPO_ROOT"
for v in product "$@"; do
  lib=$PWD/ab/lib_fw_$v.so; [ "$v" = product ] && lib=$PWD/openpystruct_amd/lib/libopenpystruct_amd.so
  echo "== $v"
  OPS_AMD_LIB=$lib python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from openpystruct_amd import frames
for bays, stories, B in ((10, 10, 16384), (12, 12, 12288), (15, 16, 12288)):
    topo = frames.grid_frame(bays, stories)
    I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda") * (1 + 0.1 * torch.rand(B, topo.Ne, dtype=torch.float64, device="cuda"))
    sol = frames.frame_solve(topo, I); torch.cuda.synchronize()
    assert int(sol.status.abs().sum()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(5):
            frames.frame_solve(topo, I, out=sol)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"{bays}x{stories} kd {topo.kd} B {B}: {best:.4f} ms  checksum {float(sol.disp.abs().sum()):.12e}")
PY
done
