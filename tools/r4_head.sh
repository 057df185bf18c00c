#!/bin/bash
# Round 4: headline levers -- buffer addressing in the column pass's memory waves (A/B), grid-ordered launch groups; pixel-MLP overlap halves
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4f
python3 -m pytest tests/test_gpu_residual.py -m gpu -x -q > gpurun_out/r4f/t_res.log 2>&1 && echo "residual tests ok" || { tail -30 gpurun_out/r4f/t_res.log; exit 1; }
bash tools/ab_bench.sh main buf0 > gpurun_out/r4f/ab_buf.txt 2>&1; cat gpurun_out/r4f/ab_buf.txt
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']/1e10,3), round(d['ms_per_step'],4))"; }
for round in 1 2; do
  run nogroup
  for g in 4 8 16 32; do NNS_BOTH_GROUP=$g run group$g; done
done > gpurun_out/r4f/groups.txt 2>&1; cat gpurun_out/r4f/groups.txt
bash tools/ab_pm.sh main ovl1 ovl2 > gpurun_out/r4f/pm_ab.txt 2>&1; cat gpurun_out/r4f/pm_ab.txt
python3 -m pytest tests/test_gpu_multirank.py -m gpu -x -q -k 'loopback or residual' > gpurun_out/r4f/t_mr.log 2>&1 && echo 'multirank ok' || tail -30 gpurun_out/r4f/t_mr.log
python3 tools/slab_loop_probe.py 2 > gpurun_out/r4f/probe.json 2> gpurun_out/r4f/probe.err && echo probe ok; tail -c 1800 gpurun_out/r4f/probe.json
echo done
