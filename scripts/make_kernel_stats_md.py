"""Markdown summary (per kernel: calls, total, average, share) of a rocprofv3 --kernel-trace --stats rocpd database.
Usage: make_kernel_stats_md.py <results.db> <out.md> "<title>" "<command>" ["extra note"]"""
import sqlite3, sys, collections
db, out, title, cmd = sys.argv[1:5]
note = sys.argv[5] if len(sys.argv) > 5 else ""
cur = sqlite3.connect(db).cursor()
agg = collections.OrderedDict()
for name, s, e in cur.execute("select name, start, end from kernels order by start"):
    d = agg.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
lines = [f"# {title}", "", f"Command: `{cmd}`, one MI355X.", ""]
if note:
    lines += [note, ""]
lines += ["| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
for name, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    lines.append(f"| `{name[:110]}` | {c} | {us:.0f} | {us / c:.1f} | {100 * us / tot:.2f} |")
lines.append("")
lines.append(f"Total kernel time {tot / 1e3:.1f} ms over {sum(v[0] for v in agg.values())} dispatches.")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
