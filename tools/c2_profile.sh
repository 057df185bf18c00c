#!/bin/bash
# Developer helper (GPU box): per-kernel GPU time of the BASELINE config 2 training iteration (rocprofv3 --kernel-trace --stats over tools/c2_run.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2 -- python3 $R/tools/c2_run.py > /tmp/c2.log 2>&1
grep "cfg2" /tmp/c2.log
f=$(ls /tmp/c2/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:8]:
    print("%-70s calls %5s avg %8.1f us total %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time per iteration (23 iterations): %.1f us" % (tot / 23 / 1e3))
PY
