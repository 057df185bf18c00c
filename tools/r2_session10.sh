#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02a
ls gpurun_out | grep r02a
cat gpurun_out/bench_r02a.json | cut -c1-600
