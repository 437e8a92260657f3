"""Print per-dispatch kernel durations of the last bench step from a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, grid_x, lds_size, vgpr_count from kernels order by start"))
idx = max(i for i, r in enumerate(rows) if "warp_entropy" in r[0])
tot = 0
for r in rows[idx:]:
    if "at::native" in r[0] or "rocclr" in r[0]: continue
    tot += (r[2]-r[1])/1e3
    print(f"{(r[2]-r[1])/1e3:9.1f} us  grid={r[3]:9d} lds={r[4]:6d} vgpr={r[5]:4d}  {r[0][:100]}")
print("sum", tot)
