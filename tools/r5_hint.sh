#!/bin/bash
# Developer helper (GPU box): the sweep-count hint -- FD tests, then a long cavity run (the solve's sweep count falls from 49 to 3) and cfg 1.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fd.py tests/test_gpu_drivers.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 100 python tools/c1_run.py
NNS_C1_FUSED=0 timeout -k 10 100 python tools/c1_run.py
timeout -k 10 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, 'neural-navier-stokes_amd')
import numpy as np, torch
from nns.chorin_fd import NavierStokesSystem
from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
for method in ('explicit', 'semi_implicit'):
    c = 64; dx = dy = 2. / (c - 1)
    u_bc = [D(0, 'left', dx, dy), D(1, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    v_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    p_bc = [D(0, 'top', dx, dy), N(0, 'bottom', dx, dy), N(0, 'left', dx, dy), N(0, 'right', dx, dy)]
    z = np.zeros((c, c))
    s = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=3000, nit=50, nx=c, ny=c, dt=1e-3, rho=1, nu=0.02, beta=1.25, method=method)
    s.simulate_device(); torch.cuda.synchronize()
    t0 = time.perf_counter(); us, vs, ps = s.simulate_device(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%s: 3000 cavity steps %.3f s = %.4f ms per step; sweeps at the last step %d; checksum %.17g' % (method, dt, dt / 3, s.sor_info()[0][0], float(us[-1].double().sum() + ps[-1].double().sum())))
PY
echo hint done
