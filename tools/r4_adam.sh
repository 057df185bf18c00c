#!/bin/bash
# Developer helper (GPU box): the one-launch Adam and the one-launch gradient zeroing -- tests, then config 2 / config 5 / physics-informed timings.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_neural.py tests/test_gpu_drivers.py -x -q -m gpu 2>&1 | tail -5 || exit 1
timeout -k 10 200 python tools/c2_run.py || exit 1
timeout -k 10 200 python tools/pinn_run.py bchw fd9 || exit 1
timeout -k 10 200 python tools/c5_time.py || exit 1
PASSES=stats bash tools/prof_any.sh r04_c2_adam tools/c2_run.py
echo adam done
