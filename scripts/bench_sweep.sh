#!/bin/bash
# usage: scripts_bench_sweep.sh <outfile>   (runs on the GPU box)
out=$1; rm -f $out
for t in 16 8 32 64; do python bench.py --steps 200 --warmup 20 --tiling $t --no-cpu-baseline --train-epochs 0 2>&1 | tail -1 >> $out; done
for t in 16 8 32; do python bench.py --batch 1048576 --steps 20 --warmup 3 --tiling $t --no-cpu-baseline --train-epochs 0 2>&1 | tail -1 >> $out; done
python - <<PY
import json
for l in open("$out"):
    try: r=json.loads(l)
    except Exception: print(l[:200]); continue
    print(r["config"]["kernel"], r["config"]["beams_per_step_per_gpu"], "%.3g solves/s"%r["value"], "%.1f us"%r["roofline"]["kernel_us"], "frac %.3f"%r["roofline"]["frac"])
PY
