#!/bin/bash
# Developer helper (GPU box): what the driver runs at round end -- the GPU tests, smoke(), and the default bench line.
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final_smoke.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; python -c "import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step']['frac'], d['cpu_baseline']['value'])"
