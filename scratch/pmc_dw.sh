#!/bin/bash
# SQ counters of the depthwise kernels on selected shapes: bash scratch/pmc_dw.sh "9,11"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DW_ONLY=$1
O=gpurun_out/pmc_dw; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python scratch/bench_dw.py > $O/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/b -- python scratch/bench_dw.py > $O/b.log 2>&1
python - <<'P'
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob("gpurun_out/pmc_dw/%s/*/*counter_collection.csv" % d)
    if not f:
        print("no csv", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60] + " g" + r["Grid_Size"]
        if "dw_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k)
        print("   " + "  ".join("%s=%.3g" % (a, b) for a, b in sorted(v.items())))
P
