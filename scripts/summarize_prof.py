#!/usr/bin/env python3
"""Summarises a gpurun_out/prof_<tag>/ directory written by scripts/profile_gpu.sh into one JSON
object: kernel-trace statistics of the beam-solve kernel plus per-launch PMC averages.
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB... (reported raw AND
converted; FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950, so both x1 and x2 are given)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
res = {"dir": d}
for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "beam_solve_kernel" in r["Name"] or "beam_rows_kernel" in r["Name"]:
            res["kernel"] = r["Name"]
            res["trace"] = {k: float(r[k]) for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev")}
pmc = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "beam_solve_kernel" in r["Kernel_Name"] or "beam_rows_kernel" in r["Kernel_Name"]:
            pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
res["dispatch"] = meta
res["pmc_avg_per_launch"] = {k: sum(v) / len(v) for k, v in sorted(pmc.items())}
res["pmc_launches"] = {k: len(v) for k, v in sorted(pmc.items())}
p = res["pmc_avg_per_launch"]
if "SQ_WAVES" in p and p["SQ_WAVES"]:
    w = p["SQ_WAVES"]
    res["per_wave"] = {
        "valu_insts": p.get("SQ_INSTS_VALU", 0) / w, "salu_insts": p.get("SQ_INSTS_SALU", 0) / w,
        "lds_insts": p.get("SQ_INSTS_LDS", 0) / w, "vmem_insts": p.get("SQ_INSTS_VMEM", 0) / w,
        "wave_cycles_x4": 4 * p.get("SQ_WAVE_CYCLES", 0) / w, "valu_active_cycles_x4": 4 * p.get("SQ_ACTIVE_INST_VALU", 0) / w,
        "wait_any_frac": p.get("SQ_WAIT_ANY", 0) / max(p.get("SQ_WAVE_CYCLES", 1), 1),
        "wait_inst_any_frac": p.get("SQ_WAIT_INST_ANY", 0) / max(p.get("SQ_WAVE_CYCLES", 1), 1),
    }
if "FETCH_SIZE" in p or "WRITE_SIZE" in p:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1 KiB... verify against the known byte count in DESIGN.md
    res["hbm"] = {"FETCH_SIZE_raw": p.get("FETCH_SIZE"), "WRITE_SIZE_raw": p.get("WRITE_SIZE")}
print(json.dumps(res, indent=1))
