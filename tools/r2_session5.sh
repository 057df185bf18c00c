#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_residual.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r2_tests5.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2_tests5.log
python tools/spec_accuracy.py > gpurun_out/r2_acc_p32.log 2>&1; cat gpurun_out/r2_acc_p32.log
NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_nop32.so python tools/spec_accuracy.py > gpurun_out/r2_acc_nop32.log 2>&1; cat gpurun_out/r2_acc_nop32.log
./ab_bench.sh main nop32 > gpurun_out/r2_ab_p32.log 2>&1; cat gpurun_out/r2_ab_p32.log
