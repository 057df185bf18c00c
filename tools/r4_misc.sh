#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4h
NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_bwdtm.so NNS_PROFILE=1 python3 tools/specbwd_run.py > gpurun_out/r4h/bwd_timing.txt 2>&1; grep -a "backward column pass" gpurun_out/r4h/bwd_timing.txt | tail -3
python3 tools/overlap_probe.py > gpurun_out/r4h/overlap1.txt 2>&1; cat gpurun_out/r4h/overlap1.txt
python3 tools/overlap_probe.py > gpurun_out/r4h/overlap2.txt 2>&1; cat gpurun_out/r4h/overlap2.txt
echo done
