"""profiles/*_pmc_traffic*.json from the two rocprofv3 --pmc passes:
python scratch/pmc_summary.py <fetch.csv> <write.csv> <out.json> <steps> [workload key model:dtype:batch:size]"""
import csv, collections, json, sys
def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        k = k.split('(')[0] if not k.startswith('_ZN') else k
        d[k][0] += 1; d[k][1] += float(r['Counter_Value'])
    return d
f, w = agg(sys.argv[1]), agg(sys.argv[2])
steps = int(sys.argv[4])
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --no-cpu-baseline --steps 2 --warmup 1` (densenet121 bf16 bs=256 320x320, %d steps incl. warm-up and the instrumented step). Counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as read. Bytes are per launch (mean over all launches of that kernel)." % steps,
       "workload": "densenet121 bf16 bs=256 320x320", "steps": steps, "kernels": {}}
if len(sys.argv) > 5:
    out["workload_key"] = sys.argv[5]
    out["workload"] = sys.argv[5]
    out["_note"] = out["_note"].replace("(densenet121 bf16 bs=256 320x320, ", "(" + sys.argv[5] + ", ")
for k in f:
    n, fv = f[k]; wn, wv = w.get(k, [0, 0.0])
    if fv * 2 + wv < 1e3: continue
    out["kernels"][k] = {"launches_per_step": n / steps, "fetch_bytes_per_launch": round(fv * 2 * 1e3 / n), "write_bytes_per_launch": round(wv * 1e3 / max(wn, 1)),
                         "hbm_bytes_per_launch": round(fv * 2 * 1e3 / n + wv * 1e3 / max(wn, 1))}
out["total_hbm_bytes_per_step"] = round(sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in out["kernels"].values()))
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print("total GB/step %.1f" % (out["total_hbm_bytes_per_step"] / 1e9))
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"])[:12]:
    print("%-50s %5.1f/step  %8.1f MB/launch  %6.2f GB/step" % (k[:50], v["launches_per_step"], v["hbm_bytes_per_launch"] / 1e6, v["hbm_bytes_per_launch"] * v["launches_per_step"] / 1e9))
