"""Per-family kernel time of one UNet forward from a rocprofv3 --kernel-trace database (dev tool).

    python tools/kernel_breakdown.py <results.db> <n_forwards>

Groups dispatches by kernel family and, for the GEMM template, by instantiation; prints ms per forward."""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel_dispatch" in t]
t = view[0]
cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t)]
ci = {c: i for i, c in enumerate(cols)}
name_c = "name" if "name" in ci else [c for c in cols if "name" in c][0]
fam, inst = {}, {}
for r in cur.execute("select * from %s" % t):
    n = r[ci[name_c]]
    d = (r[ci["end"]] - r[ci["start"]]) / 1e6
    m = re.match(r"(?:void )?(\w+)", n)
    f = m.group(1) if m else n[:40]
    a = fam.setdefault(f, [0, 0.0]); a[0] += 1; a[1] += d
    if "gemm_f16_kernel" in n:
        k = re.sub(r".*gemm_f16_kernel", "", n)[:60]
        b = inst.setdefault(k, [0, 0.0]); b[0] += 1; b[1] += d
tot = sum(a[1] for a in fam.values())
print("total %.3f ms per forward" % (tot / nf))
for k, a in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %7.1f calls/fwd %8.3f ms/fwd %5.1f%%" % (k[:44], a[0] / nf, a[1] / nf, 100 * a[1] / tot))
print("-- GEMM instantiations")
for k, a in sorted(inst.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %7.1f calls/fwd %8.3f ms/fwd" % (k, a[0] / nf, a[1] / nf))
