#!/bin/bash
# Does a HIP runtime switch cure (a) the captured hipMemsetAsync that replays garbage and (b) the generator's stalled graph launches?
for cfg in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_BLIT_KERNARG_OPT=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "AMD_DIRECT_DISPATCH=0"; do
  echo "=== ${cfg:-default}"
  ( [ -n "$cfg" ] && export $cfg; python3 -W ignore scripts/graph_reduce_probe.py 2>&1 | grep -e "^A'" -e "^C " | cut -c1-200; python3 scripts/stall_probe.py 16 2>&1 | grep -v amdgpu.ids | cut -c1-260 )
done
