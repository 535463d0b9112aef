#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in 0 10 11 12; do
  echo "order $o: $(OPS_AMD_MIX_ORDER=$o python scripts/mixed_ab.py 2560 10000 2>&1 | grep '^2560\|^10000' | tr '\n' ' ')"
done
