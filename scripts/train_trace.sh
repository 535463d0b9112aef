#!/bin/bash
# rocprofv3 kernel trace of the PINN / TFD training step; summaries -> gpurun_out/train_trace_<kind>.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for kind in pinn tfd; do
  out=gpurun_out/prof_train_$kind; rm -rf $out
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/train_epoch_bench.py --kind $kind --epochs 4 > $out.log 2>&1
  f=$(ls $out/*/*_kernel_trace.csv | head -1)
  { python3 scripts/trace_step_summary.py $f; echo ----; python3 scripts/trace_step_sequence.py $f; } > gpurun_out/train_trace_$kind.txt 2>&1
  tail -1 $out.log | cut -c1-200
done
