#!/bin/bash
cd /tmp && export TMPDIR=/tmp FRAME_BENCH_LATENCY_BATCH=0; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_frames_pmc; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/a -- python3 scripts/frame_bench2.py $1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/b -- python3 scripts/frame_bench2.py $1 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/c -- python3 scripts/frame_bench2.py $1 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_frames_pmc/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("frame_wave_kernel", "frame_tile_kernel", "frame_pack_kernel")): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
p={k:sum(v)/len(v) for k,v in acc.items()}
w=p.get("SQ_WAVES",1)
print({k:round(v/w,1) for k,v in p.items()})
print("GRBM_GUI_ACTIVE", p.get("GRBM_GUI_ACTIVE"))
PY
