"""Per-kernel summary from a rocprofv3 rocpd .db: python scratch/dbstats.py <db> <steps> [rows]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for name, n, t, a in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    nm = name.replace('(anonymous namespace)::', '').replace('void ', '')[:78]
    print("%-80s calls/step %6.1f  ms/step %7.2f  avg %8.1f us  %5.1f%%" % (nm, n / steps, t / 1e6 / steps, a / 1e3, 100 * t / tot))
print("total ms/step", tot / 1e6 / steps)
