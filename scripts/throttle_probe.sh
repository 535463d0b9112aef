#!/bin/bash
# Are the generator's stalls CFS-bandwidth throttling of the container (cpu.max quota exhausted inside a 100 ms period)?
# cgroup cpu.stat before / after the probe, CPU time and threads of the probe.
cg=/sys/fs/cgroup
echo "cpu.max: $(cat $cg/cpu.max 2>/dev/null)   affinity: $(python3 -c 'import os;print(len(os.sched_getaffinity(0)))')   nproc: $(nproc)"
echo "--- cpu.stat before"; cat $cg/cpu.stat 2>/dev/null
for cfg in "" "OMP_NUM_THREADS=1 MKL_NUM_THREADS=1" "OMP_NUM_THREADS=1 GPU_MAX_HW_QUEUES=2" ; do
  echo "=== ${cfg:-default}"
  b=$(grep -E "nr_throttled|throttled_usec" $cg/cpu.stat 2>/dev/null | tr '\n' ' ')
  ( [ -n "$cfg" ] && export $cfg; python3 scripts/stall_probe.py 16 2>&1 | grep -v amdgpu.ids | cut -c1-200; )
  a=$(grep -E "nr_throttled|throttled_usec" $cg/cpu.stat 2>/dev/null | tr '\n' ' ')
  echo "cpu.stat before: $b"; echo "cpu.stat after : $a"
done
python3 - <<'PY'
import torch, threading, os
print("torch.get_num_threads()", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
PY
