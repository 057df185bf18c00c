#!/bin/bash
# Developer helper (GPU box): GPU-side durations (rocprofv3 --kernel-trace --stats) of the pixel-MLP kernels for library variants.  usage: tools/pm_kernel_times.sh main TAG...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = main ]; then unset NNS_LIB_PATH; else export NNS_LIB_PATH=$R/ab_variants/libnns_hip_$tag.so; fi
  rm -rf /tmp/pmk_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmk_$tag -- python3 $R/tools/pm_time.py > /tmp/pmk_$tag.log 2>&1
  f=$(ls /tmp/pmk_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $tag"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pixel_mlp' in r['Name']:
        print('  %-60s calls %4s avg %8.1f us  min %8.1f us' % (r['Name'][28:88], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
