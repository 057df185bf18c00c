#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests6.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2_tests6.log
python bench_configs.py cpu_ops cfg1 cfg2 cfg5 > gpurun_out/r2_bench_configs.json 2> gpurun_out/r2_bench_configs.err; echo "configs rc=$?"; cut -c1-1500 gpurun_out/r2_bench_configs.json
python tools/c5_run.py > gpurun_out/r2_c5.log 2>&1; tail -5 gpurun_out/r2_c5.log
