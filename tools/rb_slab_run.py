"""Timing of the sharded red-black half-sweep kernel (one rank): us per sweep on 1024^2 / 4096^2, f32 and f64."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import torch
from nns import ops
out = {}
for n in (1024, 4096):
    for dt in (torch.float32, torch.float64):
        p = torch.randn(n, n, device='cuda', dtype=dt) * 0.01
        C = torch.randn(n, n, device='cuda', dtype=dt) * 0.1
        e = torch.zeros(1, device='cuda', dtype=dt)
        def sweep():
            for c in (0, 1):
                ops.fd_sor_redblack_halfsweep_(p, C, e, 0, c, 1.0 / n, 1.0 / n, 1.5)
        for _ in range(5): sweep()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): sweep()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
        b = p.element_size()
        out['n%d_%s' % (n, str(dt).split('.')[1])] = dict(us_per_sweep=1e6 * t, GBs_algorithmic=3 * n * n * b / t / 1e9)
# the whole chained solve (nns_fd_sor_redblack on a grid that does not fit LDS): 49 sweeps, tolerance never reached
for n in (1024,):
    for dt in (torch.float32, torch.float64):
        p0 = torch.randn(1, n, n, device='cuda', dtype=dt) * 0.01
        C = torch.randn(1, n, n, device='cuda', dtype=dt) * 0.1
        def solve():
            p = p0.clone()
            return ops.fd_sor_redblack_(p, C, 1.0 / n, 1.0 / n, 1.5, 5e-6, 49)
        for _ in range(3): solve()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): info = solve()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        out['solve49_n%d_%s' % (n, str(dt).split('.')[1])] = dict(ms=1e3 * t, sweeps=float(info[0, 0]))
print(json.dumps(out))
