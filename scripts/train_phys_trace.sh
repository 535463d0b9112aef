#!/bin/bash
# rocprofv3 kernel trace of the TFD + FE-residual training step; summary -> gpurun_out/train_trace_tfd_phys.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_train_phys; rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/train_phys_bench.py 3 > $out.log 2>&1
f=$(ls $out/*/*_kernel_trace.csv | head -1)
{ python3 scripts/trace_step_summary.py $f; echo ----; python3 scripts/trace_step_sequence.py $f; } > gpurun_out/train_trace_tfd_phys.txt 2>&1
tail -1 $out.log | cut -c1-200
