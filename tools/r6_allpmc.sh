#!/bin/bash
# Developer helper (GPU box): the counter passes (HBM traffic, SQ waits, LDS conflicts, instruction mix, busy) for EVERY hot kernel of the final state,
# one run script per path (tools/prof_any.sh) -> profiles/r04_all_kernels_pmc.txt via tools/prof_any_summary.py.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PASSES="stats hbm wait lds inst mem"
bash tools/prof_any.sh r04f_headline bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
bash tools/prof_any.sh r04f_mlp tools/pm_run.py
bash tools/prof_any.sh r04f_c5 tools/c5_run.py
bash tools/prof_any.sh r04f_c2 tools/c2_run.py
bash tools/prof_any.sh r04f_specbwd tools/specbwd_run.py
bash tools/prof_any.sh r04f_pinn tools/pinn_run.py bchw fd9
echo allpmc done
