"""Aggregate a rocprofv3 --kernel-trace rocpd database by kernel name over the LAST forward of the traced run.
Usage: kernel_breakdown.py <db> [marker-substring]   (marker = first kernel of a forward, default chw_to_hwc)"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
agg = collections.OrderedDict()
tot = 0.0
n = len(rows)
# last third of the trace ~ steady state; report totals divided by the number of forwards in that window if given
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = n - n // k if k > 1 else 0
for name, s, e in rows[lo:]:
    import re
    key = re.sub(r"\(anonymous namespace\)::", "", name)
    key = re.sub(r"^void ", "", key).split("(")[0][:90]
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
    tot += (e - s) / 1e3
for key, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(__import__("os").environ.get("TOPN", "30"))]:
    print(f"{us:10.1f} us  n={c:5d}  {key}")
print(f"total kernel time {tot:.1f} us over {n - lo} dispatches, wall span {(rows[-1][2]-rows[lo][1])/1e3:.1f} us")
