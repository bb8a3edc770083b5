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
big = max(runs, key=len)
busy = sum(e - s for s, e, _ in big)
span = big[-1][1] - big[0][0]
gaps = [b[0] - a[1] for a, b in zip(big, big[1:])]
nfwd = sum(1 for _, _, n in big if "timestep_embedding" in n) or 1
print("longest back-to-back run: %d kernels, %d forwards" % (len(big), nfwd))
print("per forward: kernels %.1f, busy %.3f ms, span %.3f ms, idle between kernels %.3f ms (%.2f us per launch)" % (
    len(big) / nfwd, busy / nfwd / 1e6, span / nfwd / 1e6, (span - busy) / nfwd / 1e6, (span - busy) / max(len(gaps), 1) / 1e3))
gs = sorted(gaps)
print("gap percentiles (us): p10 %.2f p50 %.2f p90 %.2f max %.2f" % (gs[len(gs) // 10] / 1e3, gs[len(gs) // 2] / 1e3, gs[9 * len(gs) // 10] / 1e3, gs[-1] / 1e3))
