"""Fold a rocprofv3 --pmc database: per kernel (name prefix + grid size) the mean of every counter and the mean dispatch
duration of the profiled pass; with GRBM_GUI_ACTIVE in the pass also the effective clock (cycles / duration).
usage: pmc_summary.py results.db [name-width]"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
W = int(sys.argv[2]) if len(sys.argv) > 2 else 70
cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
ci = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in cur.execute("select * from counters_collection"):
    key = "%s  grid=%d" % (r[ci['kernel_name']][:W], r[ci['grid_size']]) if 'grid_size' in ci else r[ci['kernel_name']][:W]
    agg[key][r[ci['counter_name']]].append(r[ci['value']])
    if 'duration' in ci and 'dispatch_id' in ci:
        dur[key][r[ci['dispatch_id']]] = r[ci['duration']]
for k, v in agg.items():
    if 'at::' in k:
        continue
    d = dur.get(k)
    dmean = (sum(d.values()) / len(d)) if d else None
    print(k + ("   dur=%.1f us (n=%d)" % (dmean / 1e3, len(d)) if dmean else ""))
    for c, vals in sorted(v.items()):
        m = sum(vals) / len(vals)
        extra = ""
        if c == "GRBM_GUI_ACTIVE" and dmean:
            extra = "   -> effective clock %.3f GHz" % (m / dmean)
        print('   %-28s n=%d mean=%.4g%s' % (c, len(vals), m, extra))
