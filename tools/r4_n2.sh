#!/bin/bash
# Developer helper (GPU box): the N > 1 bench line end to end on ONE GPU -- two ranks over gloo (device buffers staged through the host), then the
# world-1 RCCL loopback.  Rehearsals of the code path, not scaling numbers.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > gpurun_out/r04_n2_gloo.json 2> gpurun_out/r04_n2_gloo.err; echo "gloo x2 rc $?"; tail -c 600 gpurun_out/r04_n2_gloo.json
timeout -k 10 300 python bench.py --gpus 1 --mode slab --loopback --steps 10 --warmup 3 --no-secondary > gpurun_out/r04_loop_final.json 2> gpurun_out/r04_loop_final.err; echo "loopback rc $?"
timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3
echo n2 done
