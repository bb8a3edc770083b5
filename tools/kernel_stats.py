"""Export the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite database) as CSV.

    python tools/kernel_stats.py <results.db> > profiles/rNN_kernel_stats.csv

Columns: Name, Calls, TotalDurationNs (us in rocpd's view despite the name), AverageNs (us), Percentage -- the `top_kernels`
view of the database, names truncated to 160 characters."""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table') and name like '%top_kernels%'")]
w = csv.writer(sys.stdout)
if views:
    rows = list(cur.execute("select * from %s" % views[0]))
    cols = [d[0] for d in cur.description]
    w.writerow(cols)
    for r in rows:
        w.writerow([(c[:160] if isinstance(c, str) else c) for c in r])
else:   # fall back to aggregating the dispatch table
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    kd = [t for t in tabs if "kernel_dispatch" in t or t == "kernels"]
    sys.stderr.write("no top_kernels view; tables: %s\n" % tabs)
    t = kd[0]
    cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t)]
    ci = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ci else [c for c in cols if "name" in c][0]
    agg = {}
    for r in cur.execute("select * from %s" % t):
        d = r[ci["end"]] - r[ci["start"]]
        a = agg.setdefault(r[ci[name_c]], [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k[:160], a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 3)])
