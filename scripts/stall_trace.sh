#!/bin/bash
# The generator's stalled graph replays (profiles/r03_notes.md section 5) under a kernel + copy + HIP-API trace (no counters).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_stall; rm -rf $out
python3 scripts/stall_probe.py 24 > gpurun_out/r04_stall_probe_plain.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $out -- python3 scripts/stall_probe.py 24 > gpurun_out/r04_stall_probe_traced.log 2>&1
python3 scripts/stall_trace_summary.py $out > gpurun_out/r04_stall_trace_summary.txt 2>&1
rm -rf $out
cat gpurun_out/r04_stall_probe_plain.log; tail -5 gpurun_out/r04_stall_probe_traced.log; cat gpurun_out/r04_stall_trace_summary.txt
