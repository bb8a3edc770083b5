"""Busy time vs wall time of the kernel stream in a rocprofv3 kernel-trace database: sum of kernel durations, the span they
cover, and the idle time between consecutive kernels (for the graph-replayed forward: what the launches cost beyond the kernels)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
t = [x for x in tabs if x == "kernels"] or [x for x in tabs if "kernel_dispatch" in x]
cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t[0])]
nc = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = sorted((r[0], r[1], r[2]) for r in cur.execute("select start, end, %s from %s" % (nc, t[0])))
# the replays are the long back-to-back runs: split the stream where the gap exceeds 200 us
runs, cur_run = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 200000:
        runs.append(cur_run)
        cur_run = []
    cur_run.append(b)
runs.append(cur_run)
# a replayed graph runs its kernels back to back (median gap ~0); an eager stretch has microseconds between launches.  Pool every
# replay-like run of at least 200 kernels (a slow host can put > 200 us between two graph launches, which cuts the replays apart);
# without one, fall back to the longest run
def med_gap(run):
    g = sorted(b[0] - a[1] for a, b in zip(run, run[1:]))
    return g[len(g) // 2] if g else 1 << 60


replays = [r for r in runs if len(r) >= 200 and med_gap(r) < 500]
pool = replays or [max(runs, key=len)]
nk = sum(len(r) for r in pool)
busy = sum(e - s for r in pool for s, e, _ in r)
span = sum(r[-1][1] - r[0][0] for r in pool)
gaps = [b[0] - a[1] for r in pool for a, b in zip(r, r[1:])]
nfwd = sum(1 for r in pool for _, _, n in r if "timestep_embedding" in n) or 1
print("%s: %d run(s), %d kernels, %d forwards" % ("graph replays (median gap < 0.5 us)" if replays else "longest back-to-back run", len(pool), nk, nfwd))
print("per forward: kernels %.1f, busy %.3f ms, span %.3f ms, idle between kernels %.3f ms (%.2f us per launch)" % (
    nk / nfwd, busy / nfwd / 1e6, span / nfwd / 1e6, (span - busy) / nfwd / 1e6, (span - busy) / max(len(gaps), 1) / 1e3))
gs = sorted(gaps)
print("gap percentiles (us): p10 %.2f p50 %.2f p90 %.2f max %.2f" % (gs[len(gs) // 10] / 1e3, gs[len(gs) // 2] / 1e3, gs[9 * len(gs) // 10] / 1e3, gs[-1] / 1e3))
