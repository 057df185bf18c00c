#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for tag in old vB vC vD vE vF new; do
  for rep in 1 2 3 4 5; do
    if [ $tag = old ]; then unset NNS_LIB_PATH; elif [ $tag = new ]; then export NNS_LIB_PATH=$R/neural-navier-stokes_amd/csrc/libnns_hip.so; else export NNS_LIB_PATH=$R/ab_variants/libnns_hip_$tag.so; fi
    echo "$tag $(python3 ab_variants/old_tree/tools/c5_stall_probe2.py 2>&1 | grep -v amdgpu.ids | tail -1)"
  done
done
