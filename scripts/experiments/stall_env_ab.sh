#!/bin/bash
# The generator's stalled graph replays end on 100 ms boundaries of the host clock (profiles/r04_notes.md): a blocked signal wait inside
# hipGraphLaunch that only a periodic tick wakes.  Same probe with the runtime's wait policy changed.
for cfg in "" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=20000" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "=== ${cfg:-default}"
  ( [ -n "$cfg" ] && export $cfg; python3 scripts/stall_probe.py 24 2>&1 | grep -v amdgpu.ids | cut -c1-330 )
done
