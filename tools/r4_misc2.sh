#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4m
timeout -k 10 300 python3 -m pytest tests/test_gpu_residual.py -m gpu -x -q -k "capture or any_axis" > gpurun_out/r4m/t_dense.log 2>&1 && echo "dense tests ok" || tail -30 gpurun_out/r4m/t_dense.log
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --mode slab --steps 5 --warmup 2 --no-secondary > gpurun_out/r4m/gloo2.json 2> gpurun_out/r4m/gloo2.err && echo "gloo 2-rank bench ok" || tail -5 gpurun_out/r4m/gloo2.err
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --mode slab --steps 5 --warmup 2 --no-secondary --extras-timeout 1 > gpurun_out/r4m/gloo2_wd.json 2> gpurun_out/r4m/gloo2_wd.err; echo "watchdog run rc=$?"; tail -c 600 gpurun_out/r4m/gloo2_wd.json
for b in fd9 spectral; do python3 tools/pinn_run.py bchw $b 2>/dev/null | tail -1; done
echo done
