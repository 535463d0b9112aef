#!/bin/bash
# which phase of the wave-per-frame kernel costs what: rebuild with phases compiled out (results are wrong in those builds)
cd "$GRAFT_REPO_ROOT"
for flags in "" "-DFW_SKIP_BACKWARD" "-DFW_SKIP_BACKWARD -DFW_SKIP_LSTORE"; do
  echo "== flags: $flags"
  OPS_AMD_EXTRA_HIPCC_FLAGS="$flags" python -m openpystruct_amd.build --force > /dev/null 2>&1
  python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from openpystruct_amd import frames, _cabi
for bays, stories, B in ((10, 10, 16384), (15, 16, 12288)):
    topo = frames.grid_frame(bays, stories)
    I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
    sol = frames.frame_solve(topo, I); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): frames.frame_solve(topo, I, out=sol)
    e1.record(); torch.cuda.synchronize()
    print(bays, stories, B, round(e0.elapsed_time(e1) / 5, 3), "ms")
PY
done
