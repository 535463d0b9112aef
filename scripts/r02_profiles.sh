#!/bin/bash
# Round-2 evidence run (GPU box): default bench line + rocprofv3 trace / PMC passes for the three FE measurements.
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
bash scripts/profile_gpu.sh r02_hot10k > /dev/null 2>&1
PMC_STEPS=32 bash scripts/profile_gpu.sh r02_cold10k --sets 16 > /dev/null 2>&1
PMC_STEPS=4 bash scripts/profile_gpu.sh r02_sat2p20 --batch 1048576 --steps 20 > /dev/null 2>&1
for t in r02_hot10k r02_cold10k r02_sat2p20; do echo "== $t"; python3 - "$t" <<'PY'
import json,sys
r=json.load(open(f"gpurun_out/prof_{sys.argv[1]}/summary.json"))
print(r.get("kernel"), r.get("trace"), r.get("hbm"), r.get("dispatch",{}).get("Grid_Size"), r.get("per_wave"))
PY
done
tail -c 2500 gpurun_out/r02_bench_default.json
