#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5_stats -- python3 $R/tools/c5_run.py > $R/gpurun_out/c5_stats.log 2>&1
f=$(find $R/gpurun_out/c5_stats -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-220
