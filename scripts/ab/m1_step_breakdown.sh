#!/bin/bash
# kernel list of ONE steady-state M1 step (last eighth of the dispatches of a 3 + 5 step run)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_m1
timeout 300 rocprofv3 --kernel-trace -d $O/prof_m1 -o t -- python $R/bench.py --no-extras --steps 5 --cpu-sample 0 --no-pmc > $O/m1_trace.log 2>&1
cd $R
python - $(find $O/prof_m1 -name "*.db" | head -1) > $O/m1_step_breakdown.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); rows = list(db.execute("select name, start, end from kernels order by start"))
# steady-state step = from the last warp_entropy launch to the end
idx = [i for i, r in enumerate(rows) if "warp_entropy" in r[0]]
lo = idx[-1]
tot = 0.0
prev_end = rows[lo][1]
for name, s, e in rows[lo:]:
    key = re.sub(r"\(anonymous namespace\)::", "", name); key = re.sub(r"^void ", "", key).split("(")[0][:80]
    print(f"{(e - s) / 1e3:9.1f} us  gap {max(0, s - prev_end) / 1e3:7.1f} us  {key}")
    tot += (e - s) / 1e3; prev_end = e
print(f"kernels {len(rows) - lo}, kernel time {tot:.1f} us, span {(rows[-1][2] - rows[lo][1]) / 1e3:.1f} us")
PY
find $O/prof_m1 -name "*.db" -delete
