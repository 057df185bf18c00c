#!/bin/bash
# Developer helper (GPU box): what the driver runs at round end -- the GPU tests, smoke(), and the default bench line.
# A step that was killed at its time limit ends the script: no further GPU step is started after a timeout.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/final_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/final_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests hit the time limit; stopping"; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 gpurun_out/final_smoke.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "smoke hit the time limit; stopping"; exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; rc=$?; echo "bench rc=$rc"
[ $rc -eq 0 ] && python -c "import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step']['frac'], d['cpu_baseline']['value']); print({k: {kk: vv for kk, vv in v.items() if kk in ('ms', 'value', 'frac')} if isinstance(v, dict) else v for k, v in d.get('secondary', {}).items()})"
exit $rc
