#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_residual.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r2_tests15.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2_tests15.log
python tools/spec_accuracy.py 2>/dev/null | grep -A4 "fd fused\|fd standalone" | tr -d '\n' | sed 's/"n/\n"n/g' > gpurun_out/r2_acc_lap32.log; cat gpurun_out/r2_acc_lap32.log; echo
./ab_bench.sh main lap64 > gpurun_out/r2_ab_lap32.log 2>&1; cat gpurun_out/r2_ab_lap32.log
