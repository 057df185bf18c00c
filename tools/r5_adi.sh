#!/bin/bash
# Developer helper (GPU box): the LDS-resident ADI predictor -- FD tests (bitwise against the streaming kernel, goldens), drivers, multirank; cfg 1 semi-implicit timed both ways.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fd.py tests/test_gpu_drivers.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for r in 1 2; do
  echo -n "streaming "; NNS_ADI_LDS=0 NNS_C1_METHOD=semi_implicit timeout -k 10 100 python tools/c1_run.py
  echo -n "lds       "; NNS_C1_METHOD=semi_implicit timeout -k 10 100 python tools/c1_run.py
done
NNS_C1_METHOD=semi_implicit PASSES=stats bash tools/prof_any.sh r04_c1_si_lds tools/c1_run.py
echo adi done
