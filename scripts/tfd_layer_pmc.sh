#!/bin/bash
# PMC counters of the one-launch TFD kernels (separate passes, --pmc only): instruction mix, LDS conflicts, waits.  Summary -> gpurun_out/r03_tfd_layer_pmc.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_tfd_pmc; rm -rf $out
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/a -- python3 scripts/tfd_layer_trace.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $out/b -- python3 scripts/tfd_layer_trace.py > /dev/null 2>&1
python3 - $out <<'PY' > gpurun_out/r03_tfd_layer_pmc.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tfd_layer" in k:
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    w = sum(cs.get("SQ_WAVES", [0])) / max(1, len(cs.get("SQ_WAVES", [1])))
    print(k, "waves per launch", w)
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        print("   %-32s per launch %.4g   per wave %.4g" % (c, m, m / w if w else 0))
PY
cat gpurun_out/r03_tfd_layer_pmc.txt | head -50
