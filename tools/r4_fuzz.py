"""Developer helper (GPU box): random shapes through round 4's new kernels against the forms they replace.
  1. spectral residual backward (role-split column pass in the all-float32 mode) vs the composed form (standalone derivative kernels)
  2. segmented row passes (partials read from the all-to-all receive layout) vs the in-place row passes, random piece lengths / chunks
  3. the one-launch pack + halo vs the separate pack / gather kernels
  4. the fused loss head of the physics-informed step vs the same graph from tensor ops
  5. nns.optim.Adam vs torch.optim.Adam
Prints one line per case and the number of bad ones; exit status 1 if any."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns import ops
import nns.optim as nns_optim
from nns.periodic import ResidualEngine

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
NCASE = int(sys.argv[2]) if len(sys.argv) > 2 else 12
L, dt, rho = 2 * np.pi, 1e-3, 1.3
bad = 0


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def gen(seed):
    g = torch.Generator(device='cuda'); g.manual_seed(int(seed)); return g


def smooth(B, nx, ny, g, amp=1.0):
    """A few low modes plus a little noise: fields whose derivatives do not amplify float32 rounding beyond the tolerance used here."""
    x = torch.arange(nx, device='cuda', dtype=torch.float32)[None, :, None] * (2 * np.pi / nx)
    y = torch.arange(ny, device='cuda', dtype=torch.float32)[None, None, :] * (2 * np.pi / ny)
    a = torch.randn(B, 1, 1, 6, device='cuda', generator=g)
    f = a[..., 0] * torch.sin(x) * torch.cos(2 * y) + a[..., 1] * torch.cos(3 * x + y) + a[..., 2] * torch.sin(2 * x - 3 * y) + a[..., 3]
    return (amp * (f + 1e-3 * torch.randn(B, nx, ny, device='cuda', generator=g))).contiguous()


def report(what, errs, tol):
    global bad
    ok = all(np.isfinite(e) and e <= tol for e in errs)
    bad += not ok
    print('%-92s max %.1e  %s' % (what, max(errs), 'ok' if ok else 'BAD'), flush=True)


# ---- 1. spectral backward
sizes = [64, 128, 256, 512, 1024]
for it in range(NCASE):
    nx, ny = int(rng.choice(sizes)), int(rng.choice(sizes))
    B = int(rng.integers(1, max(2, min(24, int(3e7 // (nx * ny))))))
    precise = int(rng.choice([0, 0, 2]))
    nu = float(rng.choice([2 * np.pi / 1000, 1e-2]))
    g = gen(it)
    u, v = smooth(B, nx, ny, g), smooth(B, nx, ny, g)
    gr = [smooth(B, nx, ny, g) for _ in range(3)]
    Lx = L * nx / ny
    got = ops.spec_residual_bwd(u, v, *gr, dt, Lx, L, rho, nu, precise, want_prev=True)
    want = ops.spec_residual_bwd_composed(u, v, *gr, dt, Lx, L, rho, nu, precise)
    errs = [rel(a, b) for a, b in zip(got[:3], want)] + [rel(got[3], -gr[0] / dt), rel(got[4], -gr[1] / dt)]
    report('spec bwd  B %3d nx %4d ny %4d precise %d nu %.1e' % (B, nx, ny, precise, nu), errs, 2e-5)


# ---- 2. segmented row passes
def to_seg(parts, P):
    B, nloc, ny = parts[0].shape
    nyl = ny // P
    return torch.stack([torch.stack([t[:, :, s * nyl:(s + 1) * nyl] for t in parts]) for s in range(P)]).contiguous()


for it in range(NCASE):
    ny = int(rng.choice([64, 128, 256, 512, 1024]))
    P = int(rng.choice([p for p in (1, 2, 4, 8, 16) if ny // p >= max(16, ny // 16)]))
    nloc = int(rng.integers(3, 40))                                  # (a row slab needs three local rows: the C ABI refuses fewer)
    B = int(rng.integers(1, max(2, min(600, int(2e7 // (nloc * ny))))))
    precise = int(rng.choice([0, 2]))
    g = gen(100 + it)
    d = [torch.randn(B, nloc, ny, device='cuda', generator=g) for _ in range(3)]
    d += [d[0] * 0.999 + 0.001, d[1] * 0.999 - 0.001]
    parts = [torch.randn(B, nloc, ny, device='cuda', generator=g) for _ in range(3)]
    Bh = B + int(rng.integers(0, 4)); g0 = int(rng.integers(0, Bh - B + 1))
    top, bot = torch.randn(3, Bh, ny, device='cuda', generator=g), torch.randn(3, Bh, ny, device='cuda', generator=g)
    got = to_seg(parts, P)
    nu = 2 * np.pi / 1000
    rfd, rsp = ops.residual_both_rowpass_halo(*d, top, bot, [t.clone() for t in parts], dt, L / 1024, L, rho, nu, precise=precise, halo_grid0=g0)
    fd, sp = ops.residual_both_rowpass_halo_seg(*d, top, bot, got, dt, L / 1024, L, rho, nu, precise=precise, halo_grid0=g0)
    ry = ops.spec_residual_ypass_(*d, *[t.clone() for t in parts], dt, L, rho, nu, precise=precise)
    y = ops.spec_residual_ypass_seg(*d, got, dt, L, rho, nu, precise=precise)
    same = all(torch.equal(a, b) for a, b in zip(list(fd) + list(sp) + list(y), list(rfd) + list(rsp) + list(ry)))
    report('seg rows  B %3d nloc %2d ny %4d P %2d precise %d halo batch %d + %d' % (B, nloc, ny, P, precise, g0, Bh), [0.0 if same else 1.0], 0.0)

# ---- 3. pack + halo
for it in range(NCASE):
    P = int(rng.choice([1, 2, 4, 8]))
    ny = P * 4 * int(rng.integers(1, 40))
    nloc, B = int(rng.integers(1, 20)), int(rng.integers(1, 30))
    Bc = int(rng.integers(1, B + 1)); g0 = int(rng.integers(0, B - Bc + 1))
    dtype = torch.float32 if it % 3 else torch.float64
    if dtype == torch.float64 and (ny // P) % 2: ny *= 2
    g = gen(200 + it)
    fields = [torch.randn(B, nloc, ny, device='cuda', dtype=dtype, generator=g) for _ in range(3)]
    want = torch.empty(P, 3, Bc, nloc, ny // P, device='cuda', dtype=dtype)
    ops.slab_transpose_pack([t[g0:g0 + Bc].contiguous() for t in fields], want, P)
    send = torch.zeros_like(want); first = torch.zeros(3, B, ny, device='cuda', dtype=dtype); last = torch.zeros_like(first)
    ops.slab_pack_halo(fields, send, first, last, g0, P)
    same = torch.equal(send, want) and torch.equal(first, torch.stack([t[:, 0] for t in fields])) and torch.equal(last, torch.stack([t[:, -1] for t in fields]))
    report('pack+halo B %3d (chunk %d + %d) nloc %2d ny %4d P %d %s' % (B, g0, Bc, nloc, ny, P, str(dtype)[6:]), [0.0 if same else 1.0], 0.0)

# ---- 4. fused loss head
from nns.neural_spectral.physics_informed import FieldStepper, physics_informed_loss
for it in range(NCASE):
    backend = str(rng.choice(['fd5', 'fd9', 'spectral']))
    n = int(rng.choice([32, 51, 64, 100, 128, 256])) if backend != 'spectral' or it % 2 else int(rng.choice([64, 128, 256]))
    B = int(rng.integers(1, 6))
    lam, w_div = float(rng.choice([1e-3, 0.1, 1.0])), float(rng.choice([1.0, 2.5]))
    layout = 'cm' if it % 3 == 0 else 'bchw'
    g = gen(300 + it)
    state = torch.stack([smooth(B, n, n, g), smooth(B, n, n, g), smooth(B, n, n, g)], 1 if layout == 'bchw' else 0).contiguous()
    target = state + 0.01 * torch.randn(state.shape, device='cuda', generator=g)
    torch.manual_seed(it)
    model = FieldStepper(depth=int(rng.choice([2, 4, 8])), width=int(rng.choice([16, 32, 64]))).cuda()
    eng = ResidualEngine(n, n, 1e-2, rho, 0.05, 2.0, 2.0, backend=backend)
    res = []
    for fused in (True, False):
        for q in model.parameters(): q.grad = None
        t3 = physics_informed_loss(model, eng, state, target if it % 4 else None, lam=lam, w_div=w_div, layout=layout, fused=fused)
        t3[0].backward()
        res.append(([float(x.detach()) for x in t3], [q.grad.clone() for q in model.parameters()]))
    errs = [abs(a - b) / (abs(b) + 1e-30) for a, b in zip(*[r[0] for r in res]) if b != 0.0] + [rel(a, b) for a, b in zip(*[r[1] for r in res]) if float(b.norm()) > 0]
    report('pinn head %-8s n %3d B %d %s lam %.0e w_div %.1f target %d' % (backend, n, B, layout, lam, w_div, bool(it % 4)), errs, 2e-3)

# ---- 5. Adam
for it in range(max(2, NCASE // 3)):
    g = gen(400 + it)
    shapes = [tuple(int(x) for x in rng.integers(1, 70, size=int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 40)))]
    a = [torch.nn.Parameter(torch.randn(s, device='cuda', generator=g)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    kw = dict(lr=float(rng.choice([1e-3, 1e-2])), betas=(float(rng.choice([0.9, 0.5])), float(rng.choice([0.999, 0.9]))), eps=float(rng.choice([1e-8, 1e-4])),
              weight_decay=float(rng.choice([0.0, 0.1])))
    oa, ob = torch.optim.Adam(a, **kw), nns_optim.Adam(b, **kw)
    for step in range(5):
        for p, q in zip(a, b):
            p.grad = torch.randn(p.shape, device='cuda', generator=g) * 10.0 ** float(rng.integers(-4, 3)); q.grad = p.grad.clone()
        oa.step(), ob.step()
    report('adam %2d tensors %s' % (len(shapes), kw), [rel(q, p) for p, q in zip(a, b)], 3e-6)

print('%d bad' % bad)
sys.exit(1 if bad else 0)
