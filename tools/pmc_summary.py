import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols=[c[1] for c in cur.execute("pragma table_info('counters_collection')")]
ci={c:i for i,c in enumerate(cols)}
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in cur.execute("select * from counters_collection"):
    agg[r[ci['kernel_name']][:70]][r[ci['counter_name']]].append(r[ci['value']])
for k,v in agg.items():
    if 'at::' in k: continue
    print(k)
    for c,vals in sorted(v.items()):
        print('   %-28s n=%d mean=%.4g' % (c, len(vals), sum(vals)/len(vals)))
