#!/bin/bash
# A/B of the frame solve variants on the GPU box
cd "$GRAFT_REPO_ROOT"
for ws in 0 1; do for pp in 2 4; do
  echo "== FORCE_WS=$ws PP=$pp"
  OPS_AMD_FRAME_FORCE_WS=$ws OPS_AMD_FRAME_PP=$pp python scripts/frame_bench.py 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l); print(r['frame'], r['B'], round(r['ms_per_launch'],3), '%.3e'%r['frame_solves_per_s'])
    except Exception: print(l.strip()[:200])
"
done; done
