import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
# training steps actually profiled (warm-up, instrumented and timed): the loss kernel runs once per step
for r in rows:
    if 'bce_' in r['Name']:
        steps = float(r['Calls'])
        break
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    print("%-72s calls/step %6.1f  ms/step %7.2f  avg %8.1f us  %5.1f%%" % (n, int(r['Calls'])/steps, float(r['TotalDurationNs'])/1e6/steps, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print("total ms/step", tot/1e6/steps)
