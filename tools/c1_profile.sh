#!/bin/bash
# Developer helper (GPU box): per-kernel GPU time of BASELINE config 1's step (rocprofv3 --kernel-trace --stats over tools/c1_run.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1 -- python3 $R/tools/c1_run.py > /tmp/c1.log 2>&1
grep "cfg1" /tmp/c1.log
f=$(ls /tmp/c1/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:6]:
    print("%-60s calls %5s avg %8.1f us total %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("kernel time per step (400 steps): %.1f us" % (tot / 400 / 1e3))
PY
