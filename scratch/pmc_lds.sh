#!/bin/bash
# LDS / VALU activity counters of the DenseNet121 step's kernels (own --pmc passes, --kernel-trace only): bash scratch/pmc_lds.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_lds; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python bench.py --no-cpu-baseline --no-graph --steps 2 --warmup 1 > $O/a.log 2>&1 || { tail -3 $O/a.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/b -- python bench.py --no-cpu-baseline --no-graph --steps 2 --warmup 1 > $O/b.log 2>&1 || { tail -3 $O/b.log; exit 1; }
python - <<'P' | tee gpurun_out/pmc_lds/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for d in ("a", "b"):
    for f in glob.glob("gpurun_out/pmc_lds/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
rows = []
for k, v in acc.items():
    n = max(cnt[k]["GRBM_GUI_ACTIVE"], 1)
    g = v["GRBM_GUI_ACTIVE"] / n / 8          # cycles a launch is active (mean over the 8 XCDs)
    if g <= 0: continue
    per = lambda c: v.get(c, 0.0) / max(cnt[k][c], 1)
    simd = g * 1024                           # SIMD-cycles available per launch (256 CUs x 4)
    cu = g * 256
    rows.append((g * n, k, n, g, per("SQ_ACTIVE_INST_LDS") / cu, per("SQ_LDS_BANK_CONFLICT") / max(per("SQ_ACTIVE_INST_LDS"), 1), per("SQ_WAIT_INST_LDS") / max(per("SQ_WAVE_CYCLES"), 1),
                 per("SQ_ACTIVE_INST_VALU") / simd, per("SQ_INSTS_VALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM")))
print("%-44s %5s %9s  %8s %9s %9s %8s" % ("kernel (2 steps + 1 warm-up, eager)", "calls", "cycles", "LDS busy", "conflict", "wait LDS", "VALU busy"))
print("%-44s %5s %9s  %8s %9s %9s %8s" % ("", "", "/launch", "/CU cyc", "/LDS busy", "/wave cyc", "/SIMD cyc"))
for _, k, n, g, lds, conf, wl, valu, iv, il, im in sorted(rows, reverse=True)[:16]:
    print("%-44s %5d %9.0f  %7.1f%% %8.1f%% %8.1f%% %7.1f%%" % (k, n, g, 100 * lds, 100 * conf, 100 * wl, 100 * valu))
P
