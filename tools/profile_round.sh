#!/bin/bash
# Developer helper (GPU box): the evidence set for profiles/ -- bench line, rocprofv3 kernel stats of the same
# command, and the HBM-traffic counters (separate --pmc pass).  Usage: tools/profile_round.sh TAG
set -e
TAG=${1:-vX}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
[ -n "$SKIP_BENCH" ] || python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
[ -n "$SKIP_BENCH" ] || rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/stats_$TAG.log 2>&1
# FETCH_SIZE and WRITE_SIZE do not fit one pass ("exceeds the capabilities of the hardware"): one pass each
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${c}_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_${c}_$TAG.log 2>&1
done
echo profiled $TAG
