#!/bin/bash
# Developer helper (GPU box): the one-launch explicit step -- FD tests (incl. fused vs separate bitwise), drivers, then cfg 1 timed fused / separate.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fd.py tests/test_gpu_drivers.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -4 || exit 1
for r in 1 2; do
  echo -n "separate "; NNS_C1_FUSED=0 timeout -k 10 100 python tools/c1_run.py
  echo -n "fused    "; timeout -k 10 100 python tools/c1_run.py
done
PASSES=stats bash tools/prof_any.sh r04_c1_fused tools/c1_run.py
echo step done
