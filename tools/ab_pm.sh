#!/bin/bash
# same-box A/B of the pixel-MLP forward / backward (tools/pm_time.py) over library variants: usage tools/ab_pm.sh main TAG...
for round in 1 2 3; do
  for tag in "$@"; do
    if [ "$tag" = main ]; then unset NNS_LIB_PATH; else export NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_$tag.so; fi
    echo "$tag $(python tools/pm_time.py 2>/dev/null)"
  done
done
