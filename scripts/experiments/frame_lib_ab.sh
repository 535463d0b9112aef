#!/bin/bash
# A/B of frame-solve kernel variants built as separate libraries (openpystruct_amd/lib/ab_<name>.so, same C ABI; OPS_AMD_LIB selects)
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  echo "== $lib"
  FRAME_BENCH_NOCHECK=1 OPS_AMD_LIB=$PWD/openpystruct_amd/lib/ab_$lib.so python scripts/frame_bench2.py 10x10x16384 15x16x12288 5x5x32768 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l); print(r['frame'], r['B'], round(r['ms_per_launch'],3), '%.3e'%r['frame_solves_per_s'])
    except Exception: pass
"
done
