#!/bin/bash
# hot / cold / cold_stream_out / saturating of the beam solve for several tilings inside one gpurun call
# usage: scripts/r03_sweep.sh "6 16 8" [extra bench args]
tilings=${1:-"0 6 16 8"}; shift
for t in $tilings; do
  python bench.py --steps 500 --warmup 20 --tiling $t --no-cpu-baseline --train-epochs 0 "$@" 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
def f(x): return '%s %.2f us frac %.3f' % (x['kernel'] if 'kernel' in x else '', x['kernel_us'], x['frac'])
print('tiling $t hot', r['config']['kernel'], '%.2f us frac %.3f' % (r['roofline']['kernel_us'], r['roofline']['frac']))
for k in ('cold','cold_stream_out','saturating'):
    if k in r: print('tiling $t', k, f(r[k]))
"
done
