"""Per-dispatch durations (us) of the kernels whose name contains a substring, last forward of a rocprofv3 rocpd trace.
Usage: kernel_calls.py <db> <substring> [n_last]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [(n, (e - s) / 1e3) for n, s, e in cur.execute("select name, start, end from kernels order by start") if sys.argv[2] in n]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 12
for n, us in rows[-k:]:
    print(f"{us:9.1f} us  {n[:110]}")
