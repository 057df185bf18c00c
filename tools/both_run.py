"""Developer helper: FD 5-point + spectral residual, separate launches vs the fused row pass (1024^2 x 64)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns import ops
from nns.synthetic import residual_inputs
n, B = 1024, 64
f = [torch.as_tensor(np.tile(a, (B // 4, 1, 1)), device='cuda') for a in residual_inputs(4, n)]
L, dt, rho, nu = 2 * np.pi, 1e-3, 1.0, 2 * np.pi / 1000
h = L / n
o = [tuple(torch.empty_like(f[0]) for _ in range(3)) for _ in range(4)]
ops.fd_residual(*f, dt, h, h, rho, nu, 5, out=o[0]); ops.spec_residual(*f, dt, L, L, rho, nu, out=o[1])
ops.residual_both(*f, dt, L, L, rho, nu, out_fd=o[2], out_spec=o[3])
rel = lambda a, b: float((a - b).norm() / b.norm())
res = dict(spec_bitwise=all(torch.equal(a, b) for a, b in zip(o[1], o[3])), fd_rel=[rel(a, b) for a, b in zip(o[2], o[0])])
def tm(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / it
for rep in range(2):
    res['separate_ms_%d' % rep] = tm(lambda: (ops.fd_residual(*f, dt, h, h, rho, nu, 5, out=o[0]), ops.spec_residual(*f, dt, L, L, rho, nu, out=o[1])))
    res['fused_ms_%d' % rep] = tm(lambda: ops.residual_both(*f, dt, L, L, rho, nu, out_fd=o[2], out_spec=o[3]))
print(json.dumps(res))
