#!/bin/bash
# Developer helper (GPU box): the fused loss head of the physics-informed step -- its tests, then the step timed with and without it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_residual.py -x -q -m gpu -k "physics or pinn" 2>&1 | tail -5 || exit 1
for b in fd9 spectral; do
  for f in 1 0; do
    NNS_PINN_FUSED=$f timeout -k 10 200 python tools/pinn_run.py bchw $b || exit 1
  done
done
NNS_PINN_FUSED=1 timeout -k 10 200 python tools/pinn_run.py cm fd9
PASSES=stats bash tools/prof_any.sh r04_pinn_fused tools/pinn_run.py bchw fd9
echo pinn done
