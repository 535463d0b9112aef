#!/bin/bash
# epoch time of the PINN / TFD loops under variant libraries: scripts/experiments/train_lib_ab.sh ab/lib_a.so ...   (product first and last)
for lib in "" "$@" "" "$@"; do
  if [ -n "$lib" ]; then export OPS_AMD_LIB=$PWD/$lib; else unset OPS_AMD_LIB; fi
  for kind in pinn tfd; do
    echo -n "${lib:-product} $kind: "
    timeout 300 python scripts/train_epoch_bench.py --kind $kind --epochs 8 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('epoch %.3f ms, step %.1f us' % (r['value'] * 1e3, r['value'] * 1e6 / r['config']['steps_per_epoch']))
"
  done
done
