"""Summarise a rocprofv3 --pmc rocpd database: per kernel, mean counter values."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "counters_collection" in tabs:
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    q = "select kernel_name, counter_name, value from counters_collection" if "kernel_name" in cols else None
    if q is None:
        print(cols); sys.exit()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, v in cur.execute(q):
        agg[k][c].append(v)
    for k, d in agg.items():
        if "at::" in k or "rocclr" in k: continue
        print(k[:80])
        for c, vals in sorted(d.items()):
            print(f"    {c:32s} mean {sum(vals)/len(vals):.4g}  (n={len(vals)})")
else:
    print(tabs)
