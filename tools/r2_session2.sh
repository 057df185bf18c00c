#!/bin/bash
# round-2 GPU session 2: marching fused row pass -- parity, then same-box A/B against the round-1 row pass
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_residual.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r2_tests2.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2_tests2.log
tail -15 gpurun_out/r2_tests2.log
./ab_bench.sh main nomarch > gpurun_out/r2_ab_march.log 2>&1; cat gpurun_out/r2_ab_march.log
python tools/both_sizes_run.py > gpurun_out/r2_both_sizes.log 2>&1; tail -8 gpurun_out/r2_both_sizes.log
