#!/bin/bash
# rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE passes of the frames bench (GPU box, via gpurun):
#   scripts/frames_prof.sh 15x16 12288  -> gpurun_out/prof_frames_15x16/summary.json
fr=${1:-15x16}; B=${2:-12288}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_frames_$fr; rm -rf $out; mkdir -p $out
args="--workload frames --frame $fr --batch $B --steps 10 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $args > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $args > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM --output-format csv -d $out/pmc_sq -- python3 bench.py $args > $out/pmc_sq.log 2>&1
python3 - "$out" "$fr" "$B" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
d, fr, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
res = {"dir": d, "frame": fr, "frames": B, "command": f"bench.py --workload frames --frame {fr} --batch {B} --steps 10 --warmup 2"}
for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "frame_wave_kernel" in r["Name"] or "frame_pack_kernel" in r["Name"]:
            res["kernel"] = r["Name"]
            res["trace"] = {k: float(r[k]) for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev")}
pmc = defaultdict(list)
for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "frame_wave_kernel" in r["Kernel_Name"] or "frame_pack_kernel" in r["Kernel_Name"]:
            pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res["dispatch"] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Scratch_Size")}
p = {k: sum(v) / len(v) for k, v in sorted(pmc.items())}
res["pmc_avg_per_launch"] = p
res["hbm"] = {"FETCH_SIZE_raw": p.get("FETCH_SIZE"), "WRITE_SIZE_raw": p.get("WRITE_SIZE"),
              "bytes_per_launch_2xfetch_plus_write": (2.0 * p.get("FETCH_SIZE", 0) + p.get("WRITE_SIZE", 0)) * 1024.0}
if p.get("SQ_WAVES"):
    w = p["SQ_WAVES"]
    res["per_wave"] = {"valu_insts": p.get("SQ_INSTS_VALU", 0) / w, "lds_insts": p.get("SQ_INSTS_LDS", 0) / w, "vmem_insts": p.get("SQ_INSTS_VMEM", 0) / w,
                       "wave_cycles_x4": 4 * p.get("SQ_WAVE_CYCLES", 0) / w, "valu_active_cycles_x4": 4 * p.get("SQ_ACTIVE_INST_VALU", 0) / w,
                       "wait_any_frac": p.get("SQ_WAIT_ANY", 0) / max(p.get("SQ_WAVE_CYCLES", 1), 1)}
json.dump(res, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
