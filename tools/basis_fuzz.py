"""Developer helper (GPU box): random shapes through the basis loss / gradient entry points (matrix-core kernels for K <= 16 and P % 4 == 0,
packed-FMA kernels otherwise) against float64 contractions on the device."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neural-navier-stokes_amd'))
import numpy as np
import torch
from nns import ops
rng = np.random.default_rng(0)
def rel(a, b): return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))
bad = 0
for it in range(40):
    T = int(rng.integers(1, 700)); K = int(rng.integers(1, 17)); C = int(rng.integers(1, 4))
    P = int(rng.choice([4, 8, 60, 256, 1028, 4096, 4100, 10000, 65536])) if it % 4 else int(rng.integers(1, 3000))
    if T * C * P > 4e8: T = max(1, int(4e8 // (C * P)))
    g = torch.Generator(device='cuda'); g.manual_seed(it)
    coeff = torch.randn(T, K, C, device='cuda', generator=g); basis = torch.randn(K, C, P, device='cuda', generator=g); obs = torch.randn(T, C, P, device='cuda', generator=g)
    c64, b64, o64 = coeff.double(), basis.double(), obs.double()
    r = torch.einsum('tkc,kcp->tcp', c64, b64) - o64
    ss, gc, gb = ops.basis_loss_fused(coeff, basis, obs)
    e = [abs(float(ss) - float((r * r).sum())) / float((r * r).sum()), rel(gc, torch.einsum('tcp,kcp->tkc', r, b64)), rel(gb, torch.einsum('tcp,tkc->kcp', r, c64))]
    gc1, gb1 = ops.basis_loss_bwd(coeff, basis, obs, 0.37)
    e += [rel(gc1, 0.37 * torch.einsum('tcp,kcp->tkc', r, b64)), rel(gb1, 0.37 * torch.einsum('tcp,tkc->kcp', r, c64))]
    gc2, gb2 = ops.basis_expand_bwd(coeff, basis, obs)
    e += [rel(gc2, torch.einsum('tcp,kcp->tkc', o64, b64)), rel(gb2, torch.einsum('tcp,tkc->kcp', o64, c64))]
    ok = max(e) < 3e-6
    bad += not ok
    print('T %4d K %2d C %d P %6d  max err %.1e %s' % (T, K, C, P, max(e), 'ok' if ok else 'BAD ' + str(['%.1e' % x for x in e])))
print('FAILURES:', bad)
