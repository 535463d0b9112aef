#!/bin/bash
# output store policy (buffer_store aux bits: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0+sc1, 18 sc1+nt) hot vs cold at 10^4 beams
cd "$GRAFT_REPO_ROOT"
for st in 16 0 2 18 17; do
  OPS_AMD_EXTRA_HIPCC_FLAGS="-DOPS_AMD_ST=$st" python -m openpystruct_amd.build --force > /dev/null 2>&1
  python bench.py --no-cpu-baseline --train-epochs 0 --steps 200 2>/dev/null | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print('ST=$st hot', round(r['roofline']['kernel_us'],2), 'cold', round(r['cold']['kernel_us'],2), 'sat', round(r['saturating']['kernel_us'],1))"
done
