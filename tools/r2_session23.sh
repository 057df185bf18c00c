#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_neural.py tests/test_gpu_drivers.py -m gpu -q -x > gpurun_out/r2_tests23.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2_tests23.log
for r in 1 2; do python tools/pm_time.py 2>/dev/null; NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_pmuni.so python tools/pm_time.py 2>/dev/null; done | tee gpurun_out/r2_ab_pm_split.log
