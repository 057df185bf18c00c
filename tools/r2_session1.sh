#!/bin/bash
# round-2 GPU session 1: full GPU tests, A/B of row-pass XCD remap, chunked-launch experiment, slab rehearsal
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests1.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2_tests1.log
tail -5 gpurun_out/r2_tests1.log
./ab_bench.sh main xcd > gpurun_out/r2_ab_xcd.log 2>&1; cat gpurun_out/r2_ab_xcd.log
for c in 4 8 16; do
  NNS_BOTH_CHUNK=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('chunk $c', round(d['value']/1e10,3), d['ms_per_step'])" | tee -a gpurun_out/r2_chunk.log
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --backend gloo --mode slab --steps 5 --warmup 2 --batch 8 > gpurun_out/r2_slab_gloo.json 2> gpurun_out/r2_slab_gloo.err; echo "slab rc=$?"; tail -c 1500 gpurun_out/r2_slab_gloo.json
