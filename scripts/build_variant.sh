#!/bin/bash
# builds a variant of the C-ABI library for A/B runs (OPS_AMD_LIB=<path>): scripts/build_variant.sh ab/lib_x.so -DOPS_AMD_TRACE
out=$1; shift
C=openpystruct_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -ffp-contract=off -Wno-unused-command-line-argument "$@" -o $out \
  $C/beam_solve.hip $C/beam_fat.hip $C/sizing_step.hip $C/beam_residual.hip $C/frame_solve.hip $C/stencil_bn.hip $C/flat_adam.hip $C/fused_loss.hip \
  $C/beam_solve_lane.hip $C/fused_bn.hip $C/input_prep.hip $C/mlp_block.hip $C/seq_block.hip $C/seq_layer.hip $C/mem_bench.hip $C/case_draw.hip
