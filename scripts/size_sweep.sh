#!/bin/bash
# same-box A/B of tilings over batch sizes (eager launches, HIP events, two rounds): scripts/size_sweep.sh "528 16 520 8"
tilings=${1:-"528 16 520 8"}

for r in 1 2; do
for B in 10000 20000 40000 100000 300000 1048576; do
  reps=$((20000000 / B)); [ $reps -gt 400 ] && reps=400; [ $reps -lt 12 ] && reps=12
  python scripts/sat_ab.py $B "$tilings" $reps 2>/dev/null | grep "stream_out 0" | awk '{print $2, $4, $NF, $(NF-3)}' | tr '\n' ';'; echo
done; done
