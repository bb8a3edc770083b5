"""Number of UNet forwards in a rocprofv3 kernel-trace database (= launches of nhwc_to_nchw_kernel, the eps transpose that ends
every forward; round 5: the time-embedding kernels run once per sample() and no longer count forwards)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
t = [x for x in tabs if x == "kernels"] or [x for x in tabs if "kernel_dispatch" in x]
cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t[0])]
nc = "name" if "name" in cols else [c for c in cols if "name" in c][0]
print(sum(1 for r in cur.execute("select %s from %s" % (nc, t[0])) if "nhwc_to_nchw" in r[0]))
