#!/bin/bash
# PINN training step, same box: default library vs a variant (OPS_AMD_LIB), three runs each: busy us per step from the kernel trace
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for lib in "" "$1"; do
  out=gpurun_out/prof_ab_pinn; rm -rf $out
  ( [ -n "$lib" ] && export OPS_AMD_LIB=$PWD/$lib; rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/train_epoch_bench.py --kind pinn --epochs 4 > $out.log 2>&1 )
  f=$(ls $out/*/*_kernel_trace.csv | head -1)
  echo "${lib:-default}: $(python3 scripts/trace_step_summary.py $f | sed -n 2p)"
done
done
rm -rf gpurun_out/prof_ab_pinn
