"""Developer helper (GPU box): the marching fused row pass (all-float32 mode) against the separate kernels, and its timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns import ops
L, dt, rho, nu = 2 * np.pi, 1e-3, 1.3, 2 * np.pi / 1000
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def fields(B, nx, ny, seed=0):
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    x = torch.arange(nx, device='cuda', dtype=torch.float32)[:, None] * (L / nx); y = torch.arange(ny, device='cuda', dtype=torch.float32)[None, :] * (L / ny)
    base = [torch.cos(x) * torch.sin(y), -torch.sin(x) * torch.cos(y), -0.25 * (torch.cos(2 * x) + torch.cos(2 * y))]
    f = [b[None].expand(B, nx, ny) + 0.05 * torch.randn(B, nx, ny, device='cuda', generator=g) for b in base]
    f += [f[0] * 0.999 + 0.001, f[1] * 0.999 - 0.001]
    return [t.contiguous() for t in f]
for (B, nx, ny) in ((64, 1024, 1024), (64, 512, 512), (256, 256, 256), (40, 512, 1024), (1024, 128, 128), (4096, 64, 64), (2, 1024, 1024)):
    d = fields(B, nx, ny)
    fdr, spr = ops.residual_both(*d, dt, L, L, rho, nu, precise=False)
    sp = ops.spec_residual(*d, dt, L, L, rho, nu, precise=False)
    fd = ops.fd_residual(*d, dt, L / nx, L / ny, rho, nu, 5)
    fdr2, spr2 = ops.residual_both(*d, dt, L, L, rho, nu, precise=2)
    print((B, nx, ny), 'spec vs separate', ['%.1e' % rel(a, b) for a, b in zip(spr, sp)], 'fd vs standalone', ['%.1e' % rel(a, b) for a, b in zip(fdr, fd)],
          'fd vs f64-mode fused', ['%.1e' % rel(a, b) for a, b in zip(fdr, fdr2)], 'spec vs f64 mode', ['%.1e' % rel(a, b) for a, b in zip(spr, spr2)],
          'finite', all(bool(torch.isfinite(t).all()) for t in fdr + spr))
    del d, fdr, spr, sp, fd, fdr2, spr2
# slab form: local rows + halo rows from the neighbours, against the full-grid result
B, nx, ny, nl = 410, 1024, 1024, 40
d = fields(1, nx, ny, seed=3)
full_fd, full_sp = ops.residual_both(*d, dt, L, L, rho, nu, precise=False)
part = ops.spec_residual_xpass(d[0], d[1], d[2], L, rho, nu, precise=False)
r0 = 100
loc = [t[:, r0:r0 + nl].expand(B, nl, ny).contiguous() for t in d]
pl = [t[:, r0:r0 + nl].expand(B, nl, ny).contiguous() for t in part]
top = torch.stack([t[:, r0 - 1].expand(B, ny) for t in d[:3]]).contiguous(); bot = torch.stack([t[:, r0 + nl].expand(B, ny) for t in d[:3]]).contiguous()
(hf, hs) = ops.residual_both_rowpass_halo(*loc, top, bot, pl, dt, L / nx, L, rho, nu, precise=False)
print('slab', ['%.1e' % rel(a[7], b[0, r0:r0 + nl]) for a, b in zip(hf, full_fd)], ['%.1e' % rel(a[400], b[0, r0:r0 + nl]) for a, b in zip(hs, full_sp)],
      'all grids equal', all(bool((a == a[0:1]).all()) for a in hf + tuple(hs)))
del loc, pl, hf, hs
# timing at the headline shape
B, n = 64, 1024
d = fields(B, n, n)
osp = [torch.empty_like(d[0]) for _ in range(3)]; ofd = [torch.empty_like(d[0]) for _ in range(3)]
def tm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for pr in (2, False, 2, False):
    ops.spec_residual_xpass(d[0], d[1], d[2], L, rho, nu, precise=pr, out=osp)
    tr = tm(lambda: ops.residual_both(*d, dt, L, L, rho, nu, precise=pr, out_fd=ofd, out_spec=osp, rowpass_only=True))
    tx = tm(lambda: ops.spec_residual_xpass(d[0], d[1], d[2], L, rho, nu, precise=pr, out=osp))
    print('precise=%s  xpass %.3f ms  fused row pass %.3f ms' % (pr, tx, tr))
