#!/bin/bash
# Developer helper (GPU box): the round's fuzzers, a few seeds each.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for seed in 1 2 3; do
  timeout -k 10 400 python tools/r4_fuzz.py $seed 14 > gpurun_out/r4_fuzz_$seed.log 2>&1; echo "r4_fuzz seed $seed rc $? $(tail -1 gpurun_out/r4_fuzz_$seed.log)"
done
timeout -k 10 300 python tools/basis_fuzz.py > gpurun_out/basis_fuzz.log 2>&1; echo "basis_fuzz rc $? $(tail -1 gpurun_out/basis_fuzz.log)"
timeout -k 10 300 python tools/march_fuzz.py > gpurun_out/march_fuzz.log 2>&1; echo "march_fuzz rc $? $(tail -1 gpurun_out/march_fuzz.log)"
echo fuzz done
