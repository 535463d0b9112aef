#!/bin/bash
# A/B of the frame kernels on one box: row-per-lane window (TILE=0) / register tiles (TILE=1), optional variant libraries
cd "$GRAFT_REPO_ROOT"
cases="15x16x12288 10x10x16384 5x5x32768 3x3x65536"
run() { python3 scripts/frame_bench2.py $cases 2>/dev/null | python3 scripts/frame_bench_fmt.py; }
for rep in 1 2; do
echo "wave      : $(OPS_AMD_FRAME_TILE=0 run)"
echo "tile      : $(OPS_AMD_FRAME_TILE=1 run)"
for lib in "$@"; do echo "tile $lib: $(OPS_AMD_FRAME_TILE=1 OPS_AMD_LIB=$lib run)"; done
done
