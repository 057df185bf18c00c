#!/bin/bash
cd $GRAFT_REPO_ROOT
./ab_bench.sh main warm1 warm3 warm5 > gpurun_out/r2_ab_warm.log 2>&1; cat gpurun_out/r2_ab_warm.log
