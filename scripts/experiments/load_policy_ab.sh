#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for ld in 0 2 1 16; do
  OPS_AMD_EXTRA_HIPCC_FLAGS="-DOPS_AMD_LD_AUX=$ld" python -m openpystruct_amd.build --force > /dev/null 2>&1
  python bench.py --no-cpu-baseline --train-epochs 0 --steps 200 2>/dev/null | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print('LD_AUX=$ld hot', round(r['roofline']['kernel_us'],2), 'cold', round(r['cold']['kernel_us'],2), 'cold_stream', round(r['cold_stream_out']['kernel_us'],2), 'sat', round(r['saturating']['kernel_us'],1))"
done
