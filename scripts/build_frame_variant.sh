#!/bin/bash
# a variant of the library that differs in frame_solve.hip only (A/B runs, OPS_AMD_LIB=<path>): scripts/build_frame_variant.sh ab/lib_x.so -DFP_WAVES\(W\)=3
# (the other objects are the product's: python -m openpystruct_amd.build first)
out=$1; shift
O=openpystruct_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-command-line-argument "$@" -c -o ${out%.so}.frame_solve.o openpystruct_amd/csrc/frame_solve.hip || exit 1
objs=$(ls $O/*.o | grep -v frame_solve.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs ${out%.so}.frame_solve.o
