#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4c
python3 tools/slab_loop_probe.py 3 > gpurun_out/r4c/probe.json 2> gpurun_out/r4c/probe.err && echo probe ok
PASSES="stats" bash tools/prof_any.sh r4c_slab tools/slab_loop_probe.py
python3 -m pytest tests/test_gpu_neural.py -m gpu -x -q -k "pixel or mlp or bwd or backward" > gpurun_out/r4c/t_neural.log 2>&1 && echo "neural pixel tests ok" || tail -30 gpurun_out/r4c/t_neural.log
bash tools/ab_pm.sh main pmold > gpurun_out/r4c/pm_ab.txt 2>&1; cat gpurun_out/r4c/pm_ab.txt
PASSES="stats lds wait mfma mem" bash tools/prof_any.sh r4c_pm tools/mfma_run.py pm
echo r4_slab2 done
