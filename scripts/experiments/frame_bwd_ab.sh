#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for flags in "" "-DFW_BWD_DOT"; do
  echo "== flags: $flags"
  OPS_AMD_EXTRA_HIPCC_FLAGS="$flags" python -m openpystruct_amd.build --force > /dev/null 2>&1
  python -m pytest tests/test_gpu_frames.py -q -m gpu -x 2>&1 | tail -1
  python scripts/frame_bench2.py 10x10x16384 15x16x12288 5x5x32768 2>&1 | grep -v amdgpu.ids | cut -c1-160
done
