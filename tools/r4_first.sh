#!/bin/bash
# Round 4, first GPU call: this round's same-box baselines (headline line, one-GPU RCCL loopback of the slab step, spectral backward)
# and the rocprofv3 evidence VERDICT r3 asked for on the spectral backward (kernel stats + HBM / SQ counter passes).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4a
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err && echo bench ok
timeout -k 10 300 python3 bench.py --gpus 1 --mode slab --loopback --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r4a/loopback.json 2> gpurun_out/r4a/loopback.err && echo loopback ok
python3 tools/specbwd_run.py > gpurun_out/r4a/specbwd.json 2> gpurun_out/r4a/specbwd.err && echo specbwd ok
bash tools/prof_any.sh r4a_specbwd tools/specbwd_run.py
PASSES="stats lds wait mfma mem" bash tools/prof_any.sh r4a_pm tools/mfma_run.py pm
PASSES="stats lds wait mfma mem" bash tools/prof_any.sh r4a_c5 tools/mfma_run.py c5
echo r4_first done
