#!/bin/bash
# Round 4: the evidence set of the final state for profiles/ (bench line, kernel stats of the same command, HBM-traffic and SQ counter passes; the backward and the slab step)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
bash tools/profile_round.sh r04 && echo "profile_round ok"
bash tools/profile_pmc.sh && echo "pmc ok"
bash tools/prof_any.sh r04_specbwd tools/specbwd_run.py
python3 tools/specbwd_run.py > gpurun_out/r04_specbwd.json 2>/dev/null; cat gpurun_out/r04_specbwd.json
PASSES="stats" bash tools/prof_any.sh r04_slab tools/slab_loop_probe.py
for c in 1 2; do timeout -k 10 300 python3 bench.py --gpus 1 --mode slab --loopback --chunks $c --steps 20 --warmup 5 --no-secondary > gpurun_out/r04_loopback_c$c.json 2> gpurun_out/r04_loopback_c$c.err && echo "loopback chunks=$c ok"; done
python3 tools/pinn_run.py bchw fd9 2>/dev/null | tail -1; python3 tools/pinn_run.py bchw spectral 2>/dev/null | tail -1
echo evidence done
