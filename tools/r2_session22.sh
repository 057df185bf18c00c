#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_neural.py -m gpu -q -x > gpurun_out/r2_tests22.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2_tests22.log
