#!/bin/bash
# A/B of the split-row weight-gradient kernel: libraries built with other rows-per-workgroup (OPS_WG_ROWS) or with a phase compiled out
# (OPS_WG_ABLATE: 1 plain stores instead of atomics, 2 no epilogue, 3 no global loads -- results wrong in those builds)
for lib in "" FLAT scratch/lib_r03; do
  if [ "$lib" = FLAT ]; then echo -n "flat order: "; OPS_AMD_WGRAD_FLAT_ORDER=1 python3 scripts/wgrad_bench.py 2>&1 | grep -v amdgpu.ids; continue; fi
  echo -n "${lib:-default (256 rows)}: "
  ( [ -n "$lib" ] && export OPS_AMD_LIB=$PWD/$lib/libopenpystruct_amd.so; python3 scripts/wgrad_bench.py 2>&1 | grep -v amdgpu.ids )
done
