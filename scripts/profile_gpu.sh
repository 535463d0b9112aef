#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes for bench.py.
# usage: scripts/profile_gpu.sh <tag> [bench args...]      outputs -> gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
args="--no-cpu-baseline --no-extras --train-epochs 0 $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 100 --warmup 10 $args > $out/trace.log 2>&1
pmc() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps ${PMC_STEPS:-8} --warmup 2 --no-graph $args > $out/$name.log 2>&1; }
pmc pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pmc pmc_sq2 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 scripts/summarize_prof.py $out > $out/summary.json 2> $out/summary.err; cat $out/summary.json
