"""Per-kernel comparison of two rocprofv3 kernel-trace databases of the same workload (dev tool: A/B of two builds in one
gpurun call):  python tools/kernel_diff.py a.db b.db n_forwards"""
import re
import sqlite3
import sys


def load(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    t = ([x for x in tabs if x == "kernels"] or [x for x in tabs if "kernel_dispatch" in x])[0]
    cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t)]
    ci = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ci else [c for c in cols if "name" in c][0]
    agg = {}
    for r in cur.execute("select * from %s" % t):
        n = r[ci[name_c]]
        m = re.search(r"(gemm_f16_kernel|attn_fwd_kernel)<([^>]*)>", n)
        if m:
            a = [x.strip() for x in m.group(2).split(",")]
            key = "%s<%s>%s" % (m.group(1), ",".join(a[:7]), " LN" if len(a) > 8 and a[8] == "true" else "")
        else:
            m = re.search(r"(gn_\w+?_kernel(?:ILi\d+E)?|layernorm_kernel|splitk_reduce_kernel|row_stats_kernel)", n)
            key = m.group(1) if m else re.sub(r"^void ", "", n)[:48]
        d = (r[ci["end"]] - r[ci["start"]]) / 1e3
        e = agg.setdefault(key, [0, 0.0])
        e[0] += 1
        e[1] += d
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
nf = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print("total per forward: A %.3f ms  B %.3f ms" % (ta / nf / 1e3, tb / nf / 1e3))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, [0, 0])[1] + b.get(k, [0, 0])[1])):
    ca, xa = a.get(k, [0, 0.0])
    cb, xb = b.get(k, [0, 0.0])
    print("%-58s A %6.1f x %7.2f us = %7.3f ms | B %6.1f x %7.2f us = %7.3f ms | %+7.3f" % (
        k[:58], ca / nf, xa / ca if ca else 0, xa / nf / 1e3, cb / nf, xb / cb if cb else 0, xb / nf / 1e3, (xb - xa) / nf / 1e3))
