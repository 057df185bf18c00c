"""Developer helper (GPU box): 1500 steps of chorin_spectral (matrices='corrected', N = 33) with the step replayed from its HIP graph against the eager loop:
time per step, finiteness, bitwise equality of the trajectories, device-memory growth."""
import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'neural-navier-stokes_amd'))
import numpy as np, torch
from nns.chorin_spectral import NavierStokesSystem
from nns.boundary import DirichletBoundaryCondition as D
n = 33
z = np.zeros((n, n))
u_bc = [D(0., 'left', 1., 1.), D(0., 'right', 1., 1.), D(0., 'bottom', 1., 1.), D(1., 'top', 1., 1.)]
v_bc = [D(0., s, 1., 1.) for s in ('left', 'right', 'bottom', 'top')]
for mats in ('corrected',):
    s = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, nt=1500, nx=n, ny=n, dt=1e-4, nu=1.0, matrices=mats)
    m0 = torch.cuda.memory_allocated(); t0 = time.perf_counter()
    us, vs, ps = s.simulate()
    dt = time.perf_counter() - t0
    s2 = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, nt=1500, nx=n, ny=n, dt=1e-4, nu=1.0, matrices=mats)
    ue, ve, pe = s2.simulate(use_graph=False)
    print(mats, 'graph %.3f ms/step; finite %s; bitwise eager %s; |u| max %.3g; mem growth %d' % (dt / 1500 * 1e3, bool(np.isfinite(us).all() and np.isfinite(ps).all()),
          bool(np.array_equal(us, ue) and np.array_equal(ps, pe)), float(np.abs(us[-1]).max()), torch.cuda.memory_allocated() - m0))
