#!/bin/bash
# Developer helper (GPU box): the ODE row kernel on packed FMAs -- its tests, then config 2 A/B against the scalar-FMA library (same box, alternating).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_neural.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for r in 1 2 3; do
  echo -n "scalar  "; NNS_LIB_PATH=$R/ab_variants/libnns_hip_rowscalar.so timeout -k 10 100 python tools/c2_run.py
  echo -n "packed  "; timeout -k 10 100 python tools/c2_run.py
done
echo row done
