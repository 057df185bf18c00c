#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4i
python3 -m pytest tests/test_gpu_residual.py tests/test_gpu_drivers.py -m gpu -x -q -k "backward or bwd or vjp or adjoint or pinn or physics" > gpurun_out/r4i/t_bwd.log 2>&1 && echo "spectral backward tests ok" || { tail -30 gpurun_out/r4i/t_bwd.log; exit 1; }
for round in 1 2 3; do for x in 0 1; do echo "xsplit=$x $(NNS_BWD_XSPLIT=$x python3 tools/specbwd_run.py 2>/dev/null)"; done; done > gpurun_out/r4i/specbwd_ab.txt 2>&1; cat gpurun_out/r4i/specbwd_ab.txt
PASSES="stats" bash tools/prof_any.sh r4i_specbwd tools/specbwd_run.py
echo done
