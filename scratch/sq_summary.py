"""profiles/*_sq_counters*.json from one rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (+ any other SQ counters):
python scratch/sq_summary.py <counter_collection.csv> <out.json> <workload key> <steps>

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the busy counter is in cycles summed over every
SIMD (32 per v_mfma_f32_32x32x16_bf16), GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the dispatch was active
(MI355X_MICROARCH.md: 'DVFS give-back', per-instruction cycle constants)."""
import csv, collections, json, sys
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    k = k.split('(')[0] if not k.startswith('_ZN') else k
    d[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
steps = int(sys.argv[4])
out = {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU "
                "(own pass, --kernel-trace only) of `python bench.py --no-cpu-baseline --no-graph --steps 2 --warmup 1`; per-launch means; "
                "mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)", "workload_key": sys.argv[3], "steps": steps, "kernels": {}}
rows = []
for k, v in d.items():
    cnt = max(n[(k, 'GRBM_GUI_ACTIVE')], 1)
    gui = v.get('GRBM_GUI_ACTIVE', 0.0) / cnt
    mf = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(n[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')], 1)
    if gui <= 0:
        continue
    rec = {"launches_per_step": cnt / steps, "gui_active_per_launch": round(gui), "mfma_busy_cycles_per_launch": round(mf),
           "mfma_util": round(mf / (gui / 8 * 1024), 4)}
    for c in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_BUSY_CYCLES'):
        if c in v:
            rec[c.lower() + "_per_launch"] = round(v[c] / max(n[(k, c)], 1))
    out["kernels"][k] = rec
    rows.append((gui * cnt, k, rec))
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for _, k, rec in sorted(rows, reverse=True)[:14]:
    print("%-52s %6.1f/step  mfma_util %5.1f %%" % (k[:52], rec["launches_per_step"], 100 * rec["mfma_util"]))
