#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_neural.py tests/test_gpu_multirank.py tests/test_gpu_drivers.py -m gpu -q -x > gpurun_out/r2_tests21.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2_tests21.log
python bench_configs.py cfg2 cfg5 2>/dev/null | cut -c1-700
