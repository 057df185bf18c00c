#!/bin/bash
# same-box A/B: alternate variants, 3 rounds each; prints per-kernel ms.   usage: [AB_ARGS=--fast] ab_bench.sh TAG [TAG...]  ("main" = in-tree build)
for round in 1 2 3; do
  for tag in "$@"; do
    if [ "$tag" = main ]; then unset NNS_LIB_PATH; else export NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_$tag.so; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$tag', round(d['value']/1e10,3), {k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['all_kernels'].items()})"
  done
done
