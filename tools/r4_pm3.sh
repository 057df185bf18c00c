#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4l
python3 -m pytest tests/test_gpu_neural.py -m gpu -x -q -k "pixel or mlp" > gpurun_out/r4l/t_neural.log 2>&1 && echo "neural pixel tests ok" || { tail -30 gpurun_out/r4l/t_neural.log; exit 1; }
bash tools/ab_pm.sh main lay1 > gpurun_out/r4l/pm_ab.txt 2>&1; cat gpurun_out/r4l/pm_ab.txt
PASSES="stats lds wait mfma mem" bash tools/prof_any.sh r4l_pm tools/mfma_run.py pm
echo done
