"""Fold two rocprofv3 --pmc passes over the same command into the per-kernel SQ table of profiles/rNN_pmc_sq.txt:
  pass a: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
  pass b: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
usage: python tools/pmc_sq_table.py a_results.db b_results.db"""
import collections, re, sqlite3, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:64]


def fold(path):
    con = sqlite3.connect(path); cur = con.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in cur.execute("select * from counters_collection"):
        k = short(r[ci["kernel_name"]])
        acc[k][r[ci["counter_name"]]].append(r[ci["value"]])
        dur[k][r[ci["dispatch_id"]]] = r[ci["duration"]]
    return acc, dur


a, da = fold(sys.argv[1])
b, db = fold(sys.argv[2])
mean = lambda v: sum(v) / len(v) if v else float("nan")
print("%-64s %5s %9s %10s %6s %9s %8s %9s %8s %8s" % ("kernel", "n", "dur us", "wave_cyc", "active", "wait_inst", "wait_any", "mfma/wave", "lds_conf", "valu_act"))
rows = []
for k in a:
    if k.startswith("at::") or "rocclr" in k or "rocblas" in k:
        continue
    A, B = a[k], b.get(k, {})
    wc = mean(A.get("SQ_WAVE_CYCLES", []))
    if not wc or wc != wc:
        continue
    n = len(A["SQ_WAVE_CYCLES"])
    d = mean(list(da[k].values())) / 1e3
    rows.append((n * d, "%-64s %5d %9.1f %10.3e %6.2f %9.2f %8.2f %9.3f %8.3f %8.2f" % (
        k, n, d, wc, mean(A.get("SQ_ACTIVE_INST_ANY", [])) / wc, mean(A.get("SQ_WAIT_INST_ANY", [])) / wc, mean(A.get("SQ_WAIT_ANY", [])) / wc,
        mean(A.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / (4.0 * wc),
        (mean(B.get("SQ_LDS_BANK_CONFLICT", [])) / mean(B.get("SQ_LDS_IDX_ACTIVE", []))) if mean(B.get("SQ_LDS_IDX_ACTIVE", [])) else 0.0,
        (mean(B.get("SQ_ACTIVE_INST_VALU", [])) / wc) if B else float("nan"))))
for _, line in sorted(rows, reverse=True):
    print(line)
