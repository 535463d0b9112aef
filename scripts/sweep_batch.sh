#!/bin/bash
# batch-size x tiling sweep inside one gpurun call
for B in 10000 20000 40000 100000 1048576; do for t in 16 8; do
  steps=200; [ $B -ge 100000 ] && steps=40; [ $B -ge 1000000 ] && steps=20
  python bench.py --batch $B --steps $steps --warmup 5 --tiling $t --no-cpu-baseline --train-epochs 0 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print($B, $t, '%.2f us'%r['roofline']['kernel_us'], 'frac %.3f'%r['roofline']['frac'], '%.3g solves/s'%r['value'])"
done; done
