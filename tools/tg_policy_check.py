import os, sys
ROOT='/root/repo'
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')): sys.path.insert(0, p)
import numpy as np, torch
from nns import ops
from nns.synthetic import taylor_green
from oracle import periodic as OP
L, dt, rho = 2*np.pi, 1e-3, 1.0
for n in (256, 1024):
    for nu in (2*np.pi/1000, 0.0135, 0.027, 0.05):
        u, v, p = taylor_green(n, 0.1, nu, rho)
        f = [a[None].astype(np.float32) for a in (u, v, p, 0.999*u, 0.999*v)]
        d = [torch.as_tensor(a, device='cuda') for a in f]
        ref = OP.spectral_residual(*[a.astype(np.float64) for a in f], dt, L, L, rho, nu)
        out = {}
        for pr in (0, 2):
            got = ops.spec_residual(*d, dt, L, L, rho, nu, precise=pr)
            out[pr] = ['%.1e' % (np.linalg.norm(g.cpu().numpy() - r) / np.linalg.norm(r)) for g, r in zip(got[:2], ref[:2])]
        # the viscous term alone: nu * lap u, from residual differences with nu = 0
        print('n %4d nu %.4f amp %.1f  f32 %s  f64fwd %s' % (n, nu, nu*np.pi*n/(np.sqrt(3)*L), out[0], out[2]))
