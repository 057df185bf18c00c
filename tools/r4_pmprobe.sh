#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4d
python3 tools/slab_loop_probe.py 1 > gpurun_out/r4d/probe.json 2> gpurun_out/r4d/probe.err && echo probe ok
tail -c 1500 gpurun_out/r4d/probe.json
NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_tm.so python3 tools/pm_time.py > gpurun_out/r4d/pm_timing.txt 2>&1; grep -a "split backward" gpurun_out/r4d/pm_timing.txt | tail -3
bash tools/ab_pm.sh main pmold e8 e1 e32 e4 e384 e2 > gpurun_out/r4d/pm_ab.txt 2>&1; cat gpurun_out/r4d/pm_ab.txt
echo done
