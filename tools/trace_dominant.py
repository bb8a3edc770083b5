"""rocprofv3 kernel-trace average duration of one kernel (default: the halo convolution bench.py reports as dominant) ->
profiles/rNN_trace_dominant.json, stamped with the digest of the library the trace was taken with (bench.py ignores the file
when another build is loaded).

    python tools/trace_dominant.py <results.db> ["kernel name as bench.py prints it"] > profiles/r05_trace_dominant.json"""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "versatile-diffusion_amd"))
from vd_hip.loader import lib_digest

want = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_halo_kernel<256,160,32,160,512,2>"
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
t = [x for x in tabs if x == "kernels"] or [x for x in tabs if "kernel_dispatch" in x]
cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t[0])]
nc = "name" if "name" in cols else [c for c in cols if "name" in c][0]
key = re.sub(r"\s+", "", want)
# the library's name of an instantiation -> the demangled kernel name: the halo kernel carries one more template argument
# (SKIP: the folded skip convolution), printed as false / true
if key.startswith("conv3x3_halo_kernel<"):
    key = key[:-len(",skip>")] + ",true>" if key.endswith(",skip>") else key[:-1] + ",false>"
n, tot = 0, 0
for s, e, name in cur.execute("select start, end, %s from %s" % (nc, t[0])):
    if key in re.sub(r"\s+", "", name):
        n += 1
        tot += e - s
print(json.dumps({"library_digest": lib_digest(), "kernel": want, "calls": n, "avg_us": round(tot / max(n, 1) / 1e3, 3),
                  "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads"}))
