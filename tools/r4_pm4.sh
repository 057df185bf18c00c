#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4n
bash tools/ab_pm.sh main dep8 dep2 > gpurun_out/r4n/pm_ab.txt 2>&1; cat gpurun_out/r4n/pm_ab.txt
echo done
