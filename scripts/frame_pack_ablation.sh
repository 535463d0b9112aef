#!/bin/bash
# phase ablation of the packed frame kernel (GPU box): time per launch with one phase compiled out (answers are wrong: NOCHECK)
cd "$GRAFT_REPO_ROOT"; export FRAME_BENCH_LATENCY_BATCH=0 FRAME_BENCH_NOCHECK=1
for v in "" nolstore nobackward noboundary nosteps; do
  lib=$PWD/ab/lib_fp_$v.so; [ -z "$v" ] && lib=$PWD/openpystruct_amd/lib/libopenpystruct_amd.so
  [ -f $lib ] || continue
  echo "== ${v:-product}"
  OPS_AMD_LIB=$lib python scripts/frame_bench2.py "$@" 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l); print(r['frame'], r['B'], round(r['ms_per_launch'],4), '%.3e'%r['frame_solves_per_s'])
    except Exception: pass
"
done
