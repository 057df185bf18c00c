"""Developer helper (GPU box): the marching fused row pass over many batch sizes / row counts (every chunk length R = 1 .. 64, ragged last
chunks, the row-slab form with odd local row counts) against the separate kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns import ops
L, dt, rho, nu = 2 * np.pi, 1e-3, 1.3, 2 * np.pi / 1000
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def fields(B, nx, ny, seed):
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    f = [torch.randn(B, nx, ny, device='cuda', generator=g) for _ in range(3)]
    return f + [f[0] * 0.999 + 0.001, f[1] * 0.999 - 0.001]
bad = 0
for ny in (64, 256, 1024):
    for nx in (64, 256):
        for B in (1, 3, 17, 40, 100, 300, 700, 1500):
            if B * nx * ny > 6e7: continue
            d = fields(B, nx, ny, B + nx + ny)
            Lx = L * nx / ny
            fo, so = ops.residual_both(*d, dt, Lx, L, rho, nu, precise=False)
            sp = ops.spec_residual(*d, dt, Lx, L, rho, nu, precise=False)
            fd = ops.fd_residual(*d, dt, Lx / nx, L / ny, rho, nu, 5)
            e = max(max(rel(a, b) for a, b in zip(so, sp)), max(rel(a, b) for a, b in zip(fo, fd)))
            ok = e < 1e-6 and all(bool(torch.isfinite(t).all()) for t in fo + so)
            bad += not ok
            print('ny %4d nx %4d B %5d  max rel %.1e %s' % (ny, nx, B, e, 'ok' if ok else 'BAD'))
# row slabs: local row counts that are not multiples of the chunk length
for ny, nl, B in ((256, 3, 2000), (256, 5, 900), (256, 9, 333), (1024, 33, 50), (1024, 44, 400), (512, 100, 64), (64, 7, 5000)):
    nx = 4 * ((nl + 7) // 4) if nl < 16 else 2 * nl
    nx = max(64, 1 << (nx - 1).bit_length())
    d = fields(1, nx, ny, nl)
    Lx = L * nx / ny
    full_fd, full_sp = ops.residual_both(*d, dt, Lx, L, rho, nu, precise=False)
    part = ops.spec_residual_xpass(d[0], d[1], d[2], Lx, rho, nu, precise=False)
    r0 = 5
    loc = [t[:, r0:r0 + nl].expand(B, nl, ny).contiguous() for t in d]
    pl = [t[:, r0:r0 + nl].expand(B, nl, ny).contiguous() for t in part]
    top = torch.stack([t[:, r0 - 1].expand(B, ny) for t in d[:3]]).contiguous(); bot = torch.stack([t[:, r0 + nl].expand(B, ny) for t in d[:3]]).contiguous()
    hf, hs = ops.residual_both_rowpass_halo(*loc, top, bot, pl, dt, Lx / nx, L, rho, nu, precise=False)
    e = max(max(rel(a[B - 1], b[0, r0:r0 + nl]) for a, b in zip(hf, full_fd)), max(rel(a[B // 2], b[0, r0:r0 + nl]) for a, b in zip(hs, full_sp)))
    same = all(bool((a == a[0:1]).all()) for a in tuple(hf) + tuple(hs))
    ok = e < 1e-6 and same
    bad += not ok
    print('slab ny %4d nl %3d B %5d  max rel %.1e all grids equal %s %s' % (ny, nl, B, e, same, 'ok' if ok else 'BAD'))
print('FAILURES:', bad)
