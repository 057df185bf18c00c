#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_residual.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r2_tests17.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2_tests17.log
./ab_bench.sh main oldrot > gpurun_out/r2_ab_rot2.log 2>&1; cat gpurun_out/r2_ab_rot2.log
