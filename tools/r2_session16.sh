#!/bin/bash
cd $GRAFT_REPO_ROOT
./ab_bench.sh main prot > gpurun_out/r2_ab_prot2.log 2>&1; cat gpurun_out/r2_ab_prot2.log
