#!/bin/bash
cd $GRAFT_REPO_ROOT
./ab_bench.sh main skipp p32 > gpurun_out/r2_ab_pexp.log 2>&1; cat gpurun_out/r2_ab_pexp.log
python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests4.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2_tests4.log
