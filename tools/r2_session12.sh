#!/bin/bash
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']/1e10,3), {k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['all_kernels'].items()})"; }
one main; NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_sx1.so one no_transforms; NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_sx2.so one no_global
NNS_SPEC_GRID=256 one grid256; NNS_SPEC_GRID=1024 one grid1024; one main
