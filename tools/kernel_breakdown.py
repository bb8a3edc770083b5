"""Per-family kernel time of one UNet forward from a rocprofv3 --kernel-trace database or from the CSV tools/kernel_stats.py
wrote from one (dev tool).

    python tools/kernel_breakdown.py <results.db | kernel_stats.csv> <n_forwards>

Groups dispatches by kernel family and, for the GEMM and halo-conv templates, by instantiation; prints ms per forward."""
import csv
import re
import sqlite3
import sys

nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0


def rows(path):
    """(kernel name, calls, total ms)"""
    if path.endswith(".csv"):
        r = csv.reader(open(path))
        next(r)
        for row in r:
            yield row[0], int(row[1]), float(row[2]) / 1e3   # kernel_stats.py: total_duration in us
        return
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    t = ([x for x in tabs if x == "kernels"] or [x for x in tabs if "kernel_dispatch" in x])[0]
    cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t)]
    ci = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ci else [c for c in cols if "name" in c][0]
    for r in cur.execute("select * from %s" % t):
        yield r[ci[name_c]], 1, (r[ci["end"]] - r[ci["start"]]) / 1e6


def family(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?_kernel)", n)
    if m:
        return m.group(1)
    m = re.match(r"([\w:]+)", n)
    return m.group(1) if m else n[:40]


fam, inst = {}, {}
for n, c, ms in rows(sys.argv[1]):
    f = family(n)
    a = fam.setdefault(f, [0, 0.0]); a[0] += c; a[1] += ms
    m = re.search(r"(gemm_f16_kernel|conv3x3_halo_kernel|attn_fwd_kernel|xattn_kernel)(<[^>]*>)", n)
    if m:
        k = m.group(1) + re.sub(r"\s+", "", m.group(2))
        b = inst.setdefault(k, [0, 0.0]); b[0] += c; b[1] += ms
tot = sum(a[1] for a in fam.values())
print("total %.3f ms per forward" % (tot / nf))
for k, a in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %7.1f calls/fwd %8.3f ms/fwd %5.1f%%" % (k[:44], a[0] / nf, a[1] / nf, 100 * a[1] / tot))
print("-- template instantiations")
for k, a in sorted(inst.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %7.1f calls/fwd %8.3f ms/fwd" % (k, a[0] / nf, a[1] / nf))
