"""Developer helper: the reference's other small drivers' time loops alone (for rocprofv3 --kernel-trace --stats): direct_fd 50 x 50 (nit = 50) and
chorin_spectral N = 51.   usage: small_paths_run.py direct_fd|chorin_spectral"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
which = sys.argv[1] if len(sys.argv) > 1 else 'direct_fd'
if which == 'direct_fd':
    from nns.direct_fd import NavierStokesSystem
    n = 50; dx = dy = 2. / (n - 1)
    u_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'bottom', dx, dy), D(1, 'top', dx, dy)]
    v_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'bottom', dx, dy), D(0, 'top', dx, dy)]
    p_bc = [N(0, 'right', dx, dy), N(0, 'bottom', dx, dy), N(0, 'left', dx, dy), D(0, 'top', dx, dy)]
    z = np.zeros((n, n))
    s = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=200, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.1)
    s.simulate_device(); torch.cuda.synchronize(); t0 = time.perf_counter()
    s.simulate_device(); torch.cuda.synchronize()
    print('direct_fd 50x50 nit=50: %.3f ms per step' % ((time.perf_counter() - t0) / 200 * 1e3))
else:
    from nns.chorin_spectral import NavierStokesSystem
    n = 51
    z = np.zeros((n, n))
    u_bc = [D(0., 'left', 1., 1.), D(0., 'right', 1., 1.), D(0., 'bottom', 1., 1.), D(1., 'top', 1., 1.)]
    v_bc = [D(0., 'left', 1., 1.), D(0., 'right', 1., 1.), D(0., 'bottom', 1., 1.), D(0., 'top', 1., 1.)]
    s = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, nt=100, nx=n, ny=n, dt=1e-3, nu=0.1)
    s.simulate(); torch.cuda.synchronize(); t0 = time.perf_counter()
    s.simulate(); torch.cuda.synchronize()
    print('chorin_spectral N=51: %.3f ms per step' % ((time.perf_counter() - t0) / 100 * 1e3))
