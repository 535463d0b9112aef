#!/bin/bash
# PINN training step, same box, an environment switch on / off, three runs each: kernels / wall / busy us per step from the kernel trace
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for cfg in "" "$1"; do
  out=gpurun_out/prof_ab_pinn; rm -rf $out
  ( [ -n "$cfg" ] && export $cfg; rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/train_epoch_bench.py --kind ${2:-pinn} --epochs 4 > $out.log 2>&1 )
  f=$(ls $out/*/*_kernel_trace.csv | head -1)
  echo "${cfg:-default}: $(python3 scripts/trace_step_summary.py $f | sed -n 2,3p | tr '\n' ' ') | $(tail -1 $out.log | cut -c1-160)"
done
done
rm -rf gpurun_out/prof_ab_pinn
