"""Idle gaps of the GPU inside the LAST forward of a kernel trace (rocpd database): gaps > 15 us between the end of the latest kernel so far
and the start of the next one, with the kernels on either side."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:60]
# the last forward: after the last gap > 10 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - max(r[2] for r in rows[max(0, i - 40):i]) > 10_000_000: cut = i
fw = rows[cut:]
t0, t1 = fw[0][1], max(r[2] for r in fw)
busy_end, idle, gaps = fw[0][1], 0, []
for name, s, e in fw:
    if s > busy_end:
        g = s - busy_end
        idle += g
        if g > 15_000: gaps.append((g / 1e3, short(prev), short(name), (busy_end - t0) / 1e6))
    if e > busy_end: busy_end, prev = e, name
print(f"last forward: {len(fw)} kernels, span {(t1 - t0) / 1e6:.2f} ms, GPU idle {idle / 1e6:.2f} ms")
for g, a, b, at in sorted(gaps, reverse=True)[:25]:
    print(f"  {g:7.1f} us idle at +{at:6.2f} ms  after {a}  before {b}")
