#!/bin/bash
# round-2 GPU session 3: marching row pass -- band height / grid sweep (partition camping hypothesis)
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']/1e10,3), {k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['all_kernels'].items()})"; }
for R in 16 17 15 13 11 20 31 32 33; do NNS_MARCH_R=$R one "R=$R grid=512"; done 2>&1 | tee gpurun_out/r2_march_sweep.log
for R in 32 33 31; do NNS_SPEC_GRID=256 NNS_MARCH_R=$R one "R=$R grid=256"; done 2>&1 | tee -a gpurun_out/r2_march_sweep.log
for R in 8 9 7; do NNS_SPEC_GRID=1024 NNS_MARCH_R=$R one "R=$R grid=1024"; done 2>&1 | tee -a gpurun_out/r2_march_sweep.log
NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_nomarch.so one "old kernel" | tee -a gpurun_out/r2_march_sweep.log
