#!/bin/bash
# Round 4, slab step without the scatter copy: the new kernels' tests, the multi-rank tests, then the one-GPU RCCL loopback bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4b
python3 -m pytest tests/test_gpu_residual.py -m gpu -x -q -k "segmented or pack_halo or slab or row_slab or marching" > gpurun_out/r4b/t_res.log 2>&1 && echo "residual subset ok" || { tail -30 gpurun_out/r4b/t_res.log; exit 1; }
python3 -m pytest tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r4b/t_mr.log 2>&1 && echo "multirank ok" || { tail -30 gpurun_out/r4b/t_mr.log; exit 1; }
for c in 1 2; do
timeout -k 10 300 python3 bench.py --gpus 1 --mode slab --loopback --chunks $c --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r4b/loopback_c$c.json 2> gpurun_out/r4b/loopback_c$c.err && echo "loopback chunks=$c ok"
done
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --mode slab --steps 5 --warmup 2 --no-secondary > gpurun_out/r4b/gloo2.json 2> gpurun_out/r4b/gloo2.err && echo "gloo 2-rank bench ok"
echo r4_slab done
