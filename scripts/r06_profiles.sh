#!/bin/bash
# Round-6 evidence run (GPU box): default bench line + rocprofv3 trace / PMC passes for the three FE measurements + frames.
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_flags.json 2> gpurun_out/r06_bench_driver_flags.err
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
bash scripts/profile_gpu.sh r06_drv20 --steps 20 --warmup 5 > /dev/null 2>&1
bash scripts/profile_gpu.sh r06_hot10k > /dev/null 2>&1
PMC_STEPS=32 bash scripts/profile_gpu.sh r06_cold10k --sets 16 > /dev/null 2>&1
PMC_STEPS=4 bash scripts/profile_gpu.sh r06_sat2p20 --batch 1048576 --steps 20 > /dev/null 2>&1
bash scripts/frames_prof.sh 15x16 12288 > /dev/null 2>&1
bash scripts/frames_prof.sh 10x10 16384 > /dev/null 2>&1
bash scripts/frames_prof.sh 5x5 32768 > /dev/null 2>&1
bash scripts/frames_prof.sh 3x3 65536 > /dev/null 2>&1
python bench.py --workload frames --frame 15x16 > gpurun_out/r06_bench_frames_15x16.json 2>/dev/null
python bench.py --workload frames --frame 10x10 --batch 16384 > gpurun_out/r06_bench_frames_10x10.json 2>/dev/null
python bench.py --workload frames --frame 5x5 > gpurun_out/r06_bench_frames_5x5.json 2>/dev/null
python bench.py --workload frames --frame 3x3 > gpurun_out/r06_bench_frames_3x3.json 2>/dev/null
for t in r06_drv20 r06_hot10k r06_cold10k r06_sat2p20; do echo "== $t"; python3 - "$t" <<'PY'
import json,sys
r=json.load(open(f"gpurun_out/prof_{sys.argv[1]}/summary.json"))
print(r.get("kernel"), r.get("trace"), r.get("hbm"), r.get("dispatch",{}).get("Grid_Size"), r.get("per_wave"))
PY
done
for f in 15x16 10x10 5x5 3x3; do echo "== frames $f"; python3 -c "
import json; r=json.load(open('gpurun_out/prof_frames_$f/summary.json')); print(r.get('trace'), r.get('hbm'), r.get('per_wave'))"; tail -c 1500 gpurun_out/r06_bench_frames_$f.json; done
tail -c 3000 gpurun_out/r06_bench_default.json; head -c 1500 gpurun_out/r06_bench_driver_flags.json
bash scripts/train_trace.sh > gpurun_out/r06_train_trace.log 2>&1
