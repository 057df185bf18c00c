#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4e
python3 -m pytest tests/test_gpu_neural.py -m gpu -x -q -k "pixel or mlp" > gpurun_out/r4e/t_neural.log 2>&1 && echo "neural pixel tests ok" || { tail -30 gpurun_out/r4e/t_neural.log; exit 1; }
bash tools/ab_pm.sh main ovl0 > gpurun_out/r4e/pm_ab.txt 2>&1; cat gpurun_out/r4e/pm_ab.txt
echo done
