#!/bin/bash
# r03's flaky NaNs (profiles/r03_notes.md 8): framework-path TFD runs after a fast-path run in one process, explicit root gradient,
# eager tail batch.  usage: nan_hunt.sh <tag> <processes> [ENV=VALUE ...]
tag=$1; n=$2; shift 2
for kv in "$@"; do export "$kv"; done
export OPS_AMD_EXPLICIT_ROOT=${OPS_AMD_EXPLICIT_ROOT:-1} OPS_AMD_TAIL_GRAPH=${OPS_AMD_TAIL_GRAPH:-0} OPS_AMD_DEBUG_NAN=1 OPS_AMD_ADAM_ZERO=${OPS_AMD_ADAM_ZERO:-1}
for i in $(seq 1 $n); do
  python3 scripts/tfd_run_sequence.py fast,frame,frame,frame,frame 2>&1 | grep -v amdgpu.ids | cut -c1-400
done > gpurun_out/r04_nan_$tag.log 2>&1
echo "$tag: NAN lines $(grep -c -e NAN -e DEBUG_NAN -e Error gpurun_out/r04_nan_$tag.log) of $((4 * n)) framework runs"
