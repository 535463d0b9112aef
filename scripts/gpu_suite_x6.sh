#!/bin/bash
# six consecutive runs of the GPU suite on one box; log -> gpurun_out/gpu_suite_x6_late.log
cd "$GRAFT_REPO_ROOT"
log=gpurun_out/gpu_suite_x6_late.log; : > $log
fail=0
for i in 1 2 3 4 5 6; do
  echo "== run $i $(date -u +%H:%M:%S)" >> $log
  timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4 >> $log
  rc=${PIPESTATUS[0]}; echo "rc=$rc" >> $log
  [ "$rc" != "0" ] && fail=$((fail+1))
done
echo "failed runs: $fail of 6" >> $log
tail -45 $log
