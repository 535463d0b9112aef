#!/bin/bash
# copies the summaries of scripts/r06_profiles.sh from gpurun_out/ (scratch) into profiles/ (tracked)
cd /root/repo
for t in r06_drv20 r06_hot10k r06_cold10k r06_sat2p20; do
  d=gpurun_out/prof_$t
  cp $d/summary.json profiles/${t}_pmc_summary.json
  f=$(ls -S $d/trace/*/*_kernel_stats.csv | head -1); cp $f profiles/${t}_kernel_stats.csv
  [ -f $d/trace.log ] && grep -h '^{"metric"' $d/trace.log | tail -1 > profiles/${t}_bench_under_trace.json
done
for fr in 15x16 10x10 5x5 3x3; do
  d=gpurun_out/prof_frames_$fr
  cp $d/summary.json profiles/r06_frames_${fr}_pmc_summary.json
  f=$(ls -S $d/trace/*/*_kernel_stats.csv | head -1); cp $f profiles/r06_frames_${fr}_kernel_stats.csv
  cp gpurun_out/r06_bench_frames_$fr.json profiles/r06_bench_frames_$fr.json
done
cp gpurun_out/r06_bench_default.json profiles/r06_bench_default.json
grep -h '^{"metric"' gpurun_out/r06_bench_driver_flags.json | tail -1 > profiles/r06_bench_driver_flags.json
cp gpurun_out/r06_train_trace.log /dev/null 2>&1
ls -la profiles/r06_*summary.json profiles/r06_bench_default.json
