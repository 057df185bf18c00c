#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_residual.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r2_tests19.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2_tests19.log
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']/1e10,3), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['all_kernels'].items()})"; }
for r in 1 2 3; do one ws; NNS_BOTH_WS=0 one inplace; done 2>&1 | tee gpurun_out/r2_ab_ws.log
