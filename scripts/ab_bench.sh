#!/bin/bash
# A/B timing of several builds of the C-ABI library inside ONE gpurun call (same device, interleaved rounds).
# usage: scripts/ab_bench.sh "<lib1> <lib2> ..." [rounds] [bench args]
libs=$1; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do for l in $libs; do
  OPS_AMD_LIB=$PWD/$l python bench.py --steps 200 --warmup 20 --no-cpu-baseline --train-epochs 0 "$@" 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$l', r['config']['beams_per_step_per_gpu'], '%.2f us'%r['roofline']['kernel_us'], 'frac %.3f'%r['roofline']['frac'])"
done; done
