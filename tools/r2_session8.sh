#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_neural.py tests/test_gpu_multirank.py tests/test_gpu_drivers.py tests/test_gpu_chorin_spectral.py -m gpu -q -x > gpurun_out/r2_tests8.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2_tests8.log
python bench_configs.py cfg2 cfg5 > gpurun_out/r2_bench_configs2.json 2> gpurun_out/r2_bench_configs2.err; echo "configs rc=$?"; cut -c1-900 gpurun_out/r2_bench_configs2.json
