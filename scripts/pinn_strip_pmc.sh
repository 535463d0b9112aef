#!/bin/bash
# PMC counters of the PINN step's strip launches, one row per strip position in the step (separate --pmc passes): where a 7-10 us strip with ~0.4 us
# of matrix work spends its cycles.  Summary -> gpurun_out/r06_pinn_strip_pmc.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_pinn_pmc; rm -rf $out
run() { timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -- python3 scripts/train_epoch_bench.py --kind pinn --epochs 3 --cases 20000 > /dev/null 2>&1; }
n=a; run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
n=b; run SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
n=c; run GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES
python3 - $out <<'PY' > gpurun_out/r06_pinn_strip_pmc.txt
import csv, glob, sys, collections
# per pass: dispatches in order; the training step is gather, 11 strips, wgrad, adam: strip position = index among consecutive strip dispatches mod 11
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    byd = collections.defaultdict(dict)
    for r in rows:
        byd[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
        byd[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    run = 0
    for d in sorted(byd):
        r = byd[d]
        if "mlp_strip_kernel" in r["name"]:
            pos = run; run += 1
        else:
            if "mlp_gather" in r["name"]: run = 0
            continue
        if pos >= 11: continue          # evaluation passes (more strips in a row) are not the training step
        for k, v in r.items():
            if k != "name": acc[pos][k].append(v)
print("strip position in the step (0-5 forward, 6-10 backward); counters per launch (mean over the profiled training steps)")
keys = sorted({k for p in acc.values() for k in p})
for pos in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[pos].items()}
    w = c.get("SQ_WAVES", 0) or 1
    print(f"strip {pos:2d}: waves {w:6.0f}  " + "  ".join(f"{k[3:] if k.startswith('SQ_') else k} {c[k] / w:9.1f}" for k in keys if k != "SQ_WAVES" and k in c) + "   (per wave)")
PY
cat gpurun_out/r06_pinn_strip_pmc.txt | cut -c1-400
