"""GPU parity of the periodic-box Navier-Stokes residual (FD 5/9-point and Fourier-spectral
back-ends) against the CPU oracle (oracle/periodic.py), through the C ABI.

Tolerance (BASELINE.json north_star): fields within 1e-5 rel-L2 of the float64 CPU reference, for
float32 device fields.  The float64 FD kernel is held to 1e-12.  Full-size cases (1024^2, batch
64) are checked through size-independent properties: the analytic Taylor-Green answer, batch
independence (grid b of a batch == the same grid alone, bitwise) and x/y symmetry.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

NU, RHO, DT = 2 * np.pi / 1000, 1.0, 1e-3
L = 2 * np.pi
TOL = 1e-5


def dev(a, dtype=np.float32):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype), device='cuda')


def host(ts):
    return [t.cpu().numpy() for t in ts]


def inputs(batch, n, dtype=np.float32, **kw):
    from nns.synthetic import residual_inputs
    return residual_inputs(batch, n, dt=DT, nu=NU, rho=RHO, dtype=dtype, **kw)


@pytest.mark.parametrize('n', [64, 128, 256, 512, 1024])
@pytest.mark.parametrize('stencil', [5, 9])
def test_fd_residual_vs_oracle(n, stencil, gpu_device):
    from nns import ops
    from oracle import periodic as OP
    f = inputs(2, n)
    h = L / n
    got = host(ops.fd_residual(*[dev(a) for a in f], DT, h, h, RHO, NU, stencil))
    ref = OP.fd_residual(*[a.astype(np.float64) for a in f], DT, h, h, RHO, NU, stencil)
    for g, r in zip(got, ref):
        assert rel_l2(g, r) <= TOL
    got64 = host(ops.fd_residual(*[dev(a, np.float64) for a in f], DT, h, h, RHO, NU, stencil))
    for g, r in zip(got64, ref):
        assert rel_l2(g, r) <= 1e-12


@pytest.mark.parametrize('shape', [(1, 48, 200), (3, 100, 36), (2, 33, 50), (1, 7, 8)])
def test_fd_residual_ragged_sizes_both_paths(shape, gpu_device):
    """ny not a multiple of 256 / of the vector width (generic fallback kernel), nx != ny, dx != dy."""
    from nns import ops
    from oracle import periodic as OP
    rng = np.random.default_rng(7)
    f = [rng.standard_normal(shape) for _ in range(5)]
    dx, dy = 0.07, 0.11
    for stencil in (5, 9):
        ref = OP.fd_residual(*f, 0.01, dx, dy, 1.7, 0.03, stencil)
        got = host(ops.fd_residual(*[dev(a, np.float64) for a in f], 0.01, dx, dy, 1.7, 0.03, stencil))
        for g, r in zip(got, ref):
            assert rel_l2(g, r) <= 1e-12
        got = host(ops.fd_residual(*[dev(a) for a in f], 0.01, dx, dy, 1.7, 0.03, stencil))
        for g, r in zip(got, ref):
            assert rel_l2(g, r) <= TOL


@pytest.mark.parametrize('n', [64, 128, 256, 512, 1024])
@pytest.mark.parametrize('precise', [0, 1, 2])
def test_spectral_residual_vs_oracle(n, precise, gpu_device):
    """All three arithmetic policies at all five sizes, held to the north star's 1e-5: 0 = all-float32 transforms of
    forward-differenced lines (the mode the headline runs in; its viscous amplification nu pi N / (sqrt(3) L) is 1.86 at
    N = 1024 here, far under the bound of 8 the automatic policy applies), 1 = the library picks, 2 = float64 forward transforms."""
    from nns import ops
    from oracle import periodic as OP
    assert NU * np.pi * n / (np.sqrt(3) * L) <= 8
    f = inputs(2, n)
    got = host(ops.spec_residual(*[dev(a) for a in f], DT, L, L, RHO, NU, precise=precise))
    ref = OP.spectral_residual(*[a.astype(np.float64) for a in f], DT, L, L, RHO, NU)
    for g, r in zip(got, ref):
        assert rel_l2(g, r) <= TOL


@pytest.mark.parametrize('precise', [0, 1, 2])
def test_spectral_residual_non_square_and_box_lengths(precise, gpu_device):
    from nns import ops
    from oracle import periodic as OP
    rng = np.random.default_rng(8)
    nx, ny = 128, 256
    x = np.arange(nx)[:, None] / nx
    y = np.arange(ny)[None, :] / ny
    mk = lambda: (np.sin(2 * np.pi * (2 * x + rng.uniform())) * np.cos(2 * np.pi * (3 * y + rng.uniform())) +
                  0.01 * rng.standard_normal((nx, ny))).astype(np.float32)[None]
    f = [mk() for _ in range(5)]
    Lx, Ly = 1.5, 4.0
    got = host(ops.spec_residual(*[dev(a) for a in f], 2e-3, Lx, Ly, 1.2, 0.01, precise=precise))
    ref = OP.spectral_residual(*[a.astype(np.float64) for a in f], 2e-3, Lx, Ly, 1.2, 0.01)
    for g, r in zip(got, ref):
        assert rel_l2(g, r) <= TOL


def test_spectral_nyquist_and_single_modes(gpu_device):
    """Edge cases of the operator definition: the Nyquist mode is dropped by odd derivatives and kept by
    the Laplacian; single Fourier modes are differentiated exactly."""
    from nns import ops
    n = 64
    x = L * np.arange(n) / n
    X, Y = np.meshgrid(x, x, indexing='ij')
    z = np.zeros((1, n, n), np.float32)
    u = (np.cos((n // 2) * X))[None].astype(np.float32)                 # x-Nyquist mode only
    ru, rv, rd = host(ops.spec_residual(dev(u), dev(z), dev(z), dev(u), dev(z), 1.0, L, L, 1.0, 1.0))
    assert np.abs(rd).max() < 1e-4                                      # u_x of the Nyquist mode -> 0
    assert rel_l2(ru, (n // 2) ** 2 * u) < 1e-5                         # -nu lap u = +k^2 u   (u u_x = 0)
    u = (np.sin(3 * X) * np.cos(2 * Y))[None].astype(np.float32)
    ru, rv, rd = host(ops.spec_residual(dev(u), dev(z), dev(z), dev(u), dev(z), 1.0, L, L, 1.0, 0.0))
    assert rel_l2(rd, 3 * np.cos(3 * X) * np.cos(2 * Y)) < 1e-5


def test_spectral_unsupported_sizes_raise(gpu_device):
    """Axis lengths beyond both engines (FFT: powers of two in [64, 1024]; dense circulant fallback: 3 .. 2048) fail loudly, as does the
    segmented (slab) column pass on a length the FFT engine does not serve."""
    from nns import ops, _lib
    z = torch.zeros(1, 4, 4100, device='cuda')
    with pytest.raises(_lib.NnsError, match='2048'):
        ops.spec_residual(z, z, z, z, z, 1e-3, L, L, 1.0, 0.1)
    z = torch.zeros(1, 4100, 8, device='cuda')
    with pytest.raises(_lib.NnsError, match='2048'):
        ops.spec_residual(z, z, z, z, z, 1e-3, L, L, 1.0, 0.1)
    with pytest.raises(_lib.NnsError, match='2048'):
        ops.spec_residual_bwd(z, z, z, z, z, 1e-3, L, L, 1.0, 0.1)
    buf = torch.zeros(2, 3, 1, 48, 8, device='cuda')
    with pytest.raises(_lib.NnsError, match='power of two'):
        ops.spec_residual_xpass_seg(buf, buf.clone(), 1, 96, 8, 48, L, 1.0, 0.1)


@pytest.mark.parametrize('shape', [(2, 51, 51), (1, 50, 50), (2, 96, 96), (1, 96, 256), (1, 128, 100), (3, 7, 9), (1, 3, 64)])
def test_spectral_residual_any_axis_length(shape, gpu_device):
    """Axis lengths the FFT engine does not serve (the reference drivers' own 51 x 51 / 50 x 50 grids, src/chorin_fd/simulate.py:280-281,
    src/direct_fd/simulate.py:153-154; 96; odd and tiny sizes) go through the circulant-matrix form (csrc/spectral_dense.hip), per axis --
    a power-of-two axis next to an odd one keeps its FFT pass.  Forward, the fused-call surface and the backward against the float64
    oracle on rough fields, box lengths unequal."""
    from nns import ops
    from oracle import periodic as OP
    rng = np.random.default_rng(sum(shape))
    B, nx, ny = shape
    x = np.arange(nx)[:, None] / nx
    y = np.arange(ny)[None, :] / ny
    mk = lambda: (np.sin(2 * np.pi * (x + rng.uniform())) * np.cos(2 * np.pi * (y + rng.uniform())) + 0.3 * rng.standard_normal((B, nx, ny))).astype(np.float32)
    f = [mk() for _ in range(5)]
    Lx, Ly, dt, rho, nu = 1.5, 4.0, 2e-3, 1.2, 0.01
    d = [dev(a) for a in f]
    f64 = [a.astype(np.float64) for a in f]
    ref = OP.spectral_residual(*f64, dt, Lx, Ly, rho, nu)
    for g, r in zip(host(ops.spec_residual(*d, dt, Lx, Ly, rho, nu)), ref):
        assert rel_l2(g, r) <= 1e-6
    fo, so = ops.residual_both(*d, dt, Lx, Ly, rho, nu)                        # same call surface as the power-of-two sizes
    ref_fd = OP.fd_residual(*f64, dt, Lx / nx, Ly / ny, rho, nu, 5)
    for g, r in zip(host(so), ref):
        assert rel_l2(g, r) <= 1e-6
    for g, r in zip(host(fo), ref_fd):
        assert rel_l2(g, r) <= TOL
    g3 = [mk() for _ in range(3)]
    got = ops.spec_residual_bwd(d[0], d[1], *[dev(a) for a in g3], dt, Lx, Ly, rho, nu)
    refb = OP.spectral_residual_vjp(f64[0], f64[1], *[a.astype(np.float64) for a in g3], dt, Lx, Ly, rho, nu)
    for g, r in zip(host(got), refb):
        assert rel_l2(g, r) <= 1e-6


def test_spectral_residual_2048_dense(gpu_device):
    """2048 x 2048: beyond the FFT engine's 1024, inside the dense fallback's range (slow -- O(n) per point -- but exact)."""
    from nns import ops
    from oracle import periodic as OP
    f = inputs(1, 2048)
    got = host(ops.spec_residual(*[dev(a) for a in f], DT, L, L, RHO, NU))
    ref = OP.spectral_residual(*[a.astype(np.float64) for a in f], DT, L, L, RHO, NU)
    for g, r in zip(got, ref):
        assert rel_l2(g, r) <= 1e-6


def test_split_passes_equal_fused_call(gpu_device):
    from nns import ops
    f = [dev(a) for a in inputs(2, 256)]
    full = host(ops.spec_residual(*f, DT, L, L, RHO, NU))
    out = ops.spec_residual_xpass(f[0], f[1], f[2], L, RHO, NU)
    ops.spec_residual_ypass_(*f, *out, DT, L, RHO, NU)
    for a, b in zip(full, host(out)):
        np.testing.assert_array_equal(a, b)


def test_full_size_properties_1024_batch64(gpu_device):
    """BASELINE workload size (1024^2, batch 64): analytic Taylor-Green answer, batch independence,
    determinism, and FD-vs-spectral consistency at the truncation-error level."""
    from nns.periodic import ResidualEngine
    from nns.synthetic import taylor_green
    n, B = 1024, 64
    u0, v0, p0 = taylor_green(n, 0.1, NU, RHO)
    up0, vp0, _ = taylor_green(n, 0.1 - DT, NU, RHO)
    f1 = [dev(a[None]) for a in (u0, v0, p0, up0, vp0)]
    eng = ResidualEngine(n, n, DT, RHO, NU, backend='spectral')
    ru, rv, rd = host(eng(*f1))
    # exact NS solution: residual = time-discretisation error 2 nu^2 dt |u| + float32 rounding of (u-u_prev)/dt
    assert np.abs(ru).max() < 2e-4 and np.abs(rv).max() < 2e-4 and np.abs(rd).max() < 2e-5
    fd = host(eng.fd(*f1, stencil=5))
    assert np.abs(fd[0]).max() < 5e-4                                    # O(h^2) = (2 pi/1024)^2 ~ 4e-5 scale
    # batch independence + determinism on the noisy batch
    fb = [dev(a) for a in inputs(4, n)]
    big = [t.repeat(16, 1, 1).contiguous() for t in fb]                   # B = 64
    assert big[0].shape[0] == B
    for call in (eng.spectral, lambda *a: eng.fd(*a, stencil=9)):
        r64 = call(*big)
        r64b = call(*big)
        r4 = call(*fb)
        for a, b, c in zip(r64, r64b, r4):
            assert torch.equal(a, b)                                      # run-to-run bitwise
            assert torch.equal(a[:4], c) and torch.equal(a[60:], c)       # grid b of the batch == grid alone
    # the exact launch configuration of the headline (bench.py): the FUSED two-launch form at 1024^2 x 64
    bf, bs = eng.both(*big)
    bf2, bs2 = eng.both(*big)
    bf4, bs4 = eng.both(*fb)
    sep_fd, sep_sp = eng.fd(*big, stencil=5), eng.spectral(*big)
    for k in range(3):
        assert torch.equal(bf[k], bf2[k]) and torch.equal(bs[k], bs2[k])                       # run-to-run bitwise
        assert torch.equal(bf[k][:4], bf4[k]) and torch.equal(bf[k][60:], bf4[k])             # batch independence
        assert torch.equal(bs[k][:4], bs4[k]) and torch.equal(bs[k][60:], bs4[k])
        assert rel_l2(bf[k].cpu().numpy(), sep_fd[k].cpu().numpy()) < 1e-6                    # fused == separate kernels to rounding
        assert rel_l2(bs[k].cpu().numpy(), sep_sp[k].cpu().numpy()) < 1e-6
    from oracle import periodic as OP
    f64 = [a.astype(np.float64) for a in inputs(4, n)]
    ref_fd, ref_sp = OP.fd_residual(*f64, DT, L / n, L / n, RHO, NU, 5), OP.spectral_residual(*f64, DT, L, L, RHO, NU)
    for k in range(3):
        assert rel_l2(bf[k][60:].cpu().numpy(), ref_fd[k]) <= TOL and rel_l2(bs[k][60:].cpu().numpy(), ref_sp[k]) <= TOL


def test_slab_residual_single_rank_uses_hip_backend(gpu_device):
    """nns.slab.SlabResidual with its default (HIP) compute back-end on one rank: the P == 1 path does the
    same packing / padding / transposes locally, so the result must equal the direct ops."""
    import torch.distributed as dist
    from nns import ops
    from nns.slab import SlabResidual
    created = False
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29917', rank=0, world_size=1)
        created = True
    try:
        f = [dev(a) for a in inputs(3, 128)]
        s = SlabResidual(128, 128, DT, RHO, NU, L, L)
        h = L / 128
        for a, b in zip(s.fd(*f, stencil=9), ops.fd_residual(*f, DT, h, h, RHO, NU, 9)):
            assert torch.equal(a, b)
        for a, b in zip(s.spectral(*f), ops.spec_residual(*f, DT, L, L, RHO, NU)):
            assert torch.equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize('shape', [(2, 64, 128), (1, 256, 64), (1, 1024, 1024)])
def test_rfft2_irfft2_match_numpy(shape, gpu_device):
    from nns import ops
    rng = np.random.default_rng(12)
    f = rng.standard_normal(shape).astype(np.float32)
    spec = ops.spec_rfft2(dev(f))
    ref = np.fft.rfft2(f.astype(np.float64))
    assert spec.shape == ref.shape and spec.dtype == torch.complex64
    crel = lambda a, b: np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())          # complex-aware
    assert crel(spec.cpu().numpy().astype(np.complex128), ref) < 2e-6
    back = ops.spec_irfft2(spec, shape[2])
    assert rel_l2(back.cpu().numpy(), f) < 2e-6
    back2 = ops.spec_irfft2(torch.as_tensor(ref.astype(np.complex64), device='cuda'), shape[2])   # from numpy's spectrum
    assert rel_l2(back2.cpu().numpy(), np.fft.irfft2(ref, s=shape[1:])) < 2e-6


@pytest.mark.parametrize('n', [64, 256, 1024])
def test_spec_derivs_vs_oracle(n, gpu_device):
    from nns import ops
    from oracle import periodic as OP
    f = inputs(2, n)[0]
    d = ops.spec_derivs(dev(f), L, 2 * L)
    fx, fy, lap = OP.spectral_derivs(f.astype(np.float64), L, 2 * L)
    assert rel_l2(d['x'].cpu().numpy(), fx) < TOL and rel_l2(d['y'].cpu().numpy(), fy) < TOL and rel_l2(d['lap'].cpu().numpy(), lap) < TOL
    only = ops.spec_derivs(dev(f), L, 2 * L, want=('y',))
    assert list(only) == ['y'] and torch.equal(only['y'], d['y'])


@pytest.mark.parametrize("stencil", [5, 9])
@pytest.mark.parametrize("shape,dtype", [((2, 64, 128), 'float32'), ((1, 37, 30), 'float32'), ((2, 48, 64), 'float64'), ((1, 5, 7), 'float64')])
def test_fd_residual_backward_vs_oracle(gpu_device, stencil, shape, dtype):
    """Adjoint-stencil kernel (vectorised and ragged-size paths) vs the oracle VJP: 1e-5 in float32, 1e-12 in float64."""
    from nns import ops
    from oracle import periodic as OP
    rng = np.random.default_rng(21)
    B, nx, ny = shape
    Lx, Ly, dt, rho, nu = 2 * np.pi, 3.0, 1e-2, 1.2, 0.03
    dx, dy = Lx / nx, Ly / ny
    f = [rng.standard_normal(shape).astype(dtype) for _ in range(5)]          # u, v, g_u, g_v, g_div
    ref = OP.fd_residual_vjp(*[a.astype(np.float64) for a in f], dt, dx, dy, rho, nu, stencil)
    got = ops.fd_residual_bwd(*[torch.as_tensor(a, device='cuda') for a in f], dt, dx, dy, rho, nu, stencil)
    tol = 1e-5 if dtype == 'float32' else 1e-12
    for name, g, r in zip(('u', 'v', 'p', 'u_prev', 'v_prev'), got, ref):
        assert rel_l2(g.cpu().numpy(), r) < tol, (name, stencil, shape, dtype)
    gu, gv, gp, a, b = ops.fd_residual_bwd(*[torch.as_tensor(x, device='cuda') for x in f], dt, dx, dy, rho, nu, stencil, want_prev=False)
    assert a is None and b is None and rel_l2(gp.cpu().numpy(), ref[2]) < tol


def test_physics_loss_autograd(gpu_device):
    """The differentiable residual: loss.backward() through the HIP forward/backward pair equals the oracle's
    gradient of the same mean-square residual loss (chain rule by hand), and gradient descent on (u, v, p) lowers it."""
    from nns.periodic import ResidualEngine
    from oracle import periodic as OP
    rng = np.random.default_rng(4)
    B, n = 2, 64
    dt, rho, nu = 1e-2, 1.0, 0.05
    eng = ResidualEngine(n, n, dt, rho, nu, backend='fd9')
    f = [rng.standard_normal((B, n, n)) * 0.3 for _ in range(5)]
    t = [torch.as_tensor(a, dtype=torch.float64, device='cuda').requires_grad_(True) for a in f]
    loss = eng.physics_loss(*t, w_div=2.0)
    loss.backward()
    r = OP.fd_residual(*f, dt, eng.dx, eng.dy, rho, nu, 9)
    N = B * n * n
    g = (2 * r[0] / N, 2 * r[1] / N, 2 * 2.0 * r[2] / N)
    ref = OP.fd_residual_vjp(f[0], f[1], *g, dt, eng.dx, eng.dy, rho, nu, 9)
    assert abs(loss.item() - ((r[0]**2).mean() + (r[1]**2).mean() + 2.0 * (r[2]**2).mean())) < 1e-10 * loss.item()
    for tt, rr in zip(t, ref):
        assert rel_l2(tt.grad.cpu().numpy(), rr) < 1e-11


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 256, 64), (1, 1024, 512), (3, 512, 1024)])
def test_spectral_residual_backward_vs_oracle(gpu_device, shape):
    """VJP of the spectral residual (adjoint operators applied by the HIP spectral engine) vs the oracle, 1e-5; and
    through autograd on band-limited fields."""
    from nns import ops
    from nns.periodic import ResidualEngine
    from nns.synthetic import residual_inputs
    from oracle import periodic as OP
    B, nx, ny = shape
    Lx, Ly, dt, rho, nu = 2 * np.pi, 2 * np.pi, 1e-2, 1.1, 0.02
    rng = np.random.default_rng(8)
    def smooth():
        f = rng.standard_normal(shape)
        F = np.fft.rfft2(f)
        kx, ky = np.abs(np.fft.fftfreq(nx, 1.0 / nx))[:, None], np.fft.rfftfreq(ny, 1.0 / ny)[None, :]
        return np.fft.irfft2(F * np.exp(-(kx**2 + ky**2) / 60.0), s=(nx, ny)).astype(np.float32)
    f = [smooth() for _ in range(5)]
    f = [a / np.abs(a).max() for a in f]
    ref = OP.spectral_residual_vjp(*[a.astype(np.float64) for a in f], dt, Lx, Ly, rho, nu)
    got = ops.spec_residual_bwd(*[torch.as_tensor(a, device='cuda') for a in f], dt, Lx, Ly, rho, nu)
    for name, g, r in zip(('u', 'v', 'p', 'u_prev', 'v_prev'), got, ref):
        assert rel_l2(g.cpu().numpy(), r) < 1e-5, (name, shape)
    comp = ops.spec_residual_bwd_composed(*[torch.as_tensor(a, device='cuda') for a in f], dt, Lx, Ly, rho, nu)
    for g, r in zip(comp, ref):                                  # the independent composition agrees too
        assert rel_l2(g.cpu().numpy(), r) < 1e-5
    no_prev = ops.spec_residual_bwd(*[torch.as_tensor(a, device='cuda') for a in f], dt, Lx, Ly, rho, nu, want_prev=False)
    assert no_prev[3] is None and rel_l2(no_prev[0].cpu().numpy(), ref[0]) < 1e-5
    eng = ResidualEngine(nx, ny, dt, rho, nu, Lx, Ly, backend='spectral')
    t = [torch.as_tensor(a, device='cuda').requires_grad_(True) for a in f]
    loss = eng.physics_loss(*t)
    loss.backward()
    r = OP.spectral_residual(*[a.astype(np.float64) for a in f], dt, Lx, Ly, rho, nu)
    N = B * nx * ny
    ref = OP.spectral_residual_vjp(f[0].astype(np.float64), f[1].astype(np.float64), 2 * r[0] / N, 2 * r[1] / N, 2 * r[2] / N, dt, Lx, Ly, rho, nu)
    for tt, rr in zip(t, ref):
        assert rel_l2(tt.grad.cpu().numpy(), rr) < 2e-5


@pytest.mark.parametrize("backend", ["fd5", "fd9", "spectral"])
def test_backward_adjoint_identity_full_size(gpu_device, backend):
    """Size-independent property at BASELINE's grid size (1024^2): <J d, g> = <d, J^T g> with both sides from the HIP
    kernels.  The residual is quadratic in (u, v) and linear in the rest, so J d = (r(w + d) - r(w - d)) / 2 exactly up
    to float32 rounding; sums are accumulated in float64."""
    from nns.periodic import ResidualEngine
    from nns.synthetic import residual_inputs
    n, B = 1024, 2
    eng = ResidualEngine(n, n, 1e-3, 1.0, 2 * np.pi / 1000, backend=backend)
    w = [torch.as_tensor(a, device='cuda') for a in residual_inputs(B, n)]
    g = torch.Generator(device='cuda').manual_seed(3)
    def smooth():
        f = torch.randn(B, n, n, device='cuda', generator=g)
        F = torch.fft.rfft2(f)
        kx = torch.fft.fftfreq(n, 1.0 / n, device='cuda').abs()[:, None]; ky = torch.fft.rfftfreq(n, 1.0 / n, device='cuda')[None, :]
        s = torch.fft.irfft2(F * torch.exp(-(kx**2 + ky**2) / 200.0), s=(n, n))
        return (s / s.abs().max()).contiguous()
    d = [0.05 * smooth() for _ in range(5)]
    gr = [smooth() for _ in range(3)]
    rp = eng(*[a + b for a, b in zip(w, d)])
    rp = [t.clone() for t in rp]
    rm = eng(*[a - b for a, b in zip(w, d)])
    Jd = [(a.double() - b.double()) / 2 for a, b in zip(rp, rm)]
    lhs = sum((a * b.double()).sum().item() for a, b in zip(Jd, gr))
    if backend == 'spectral':
        from nns import ops
        grads = ops.spec_residual_bwd(w[0], w[1], *gr, eng.dt, eng.Lx, eng.Ly, eng.rho, eng.nu)
    else:
        from nns import ops
        grads = ops.fd_residual_bwd(w[0], w[1], *gr, eng.dt, eng.dx, eng.dy, eng.rho, eng.nu, 5 if backend == 'fd5' else 9)
    rhs = sum((a.double() * b.double()).sum().item() for a, b in zip(grads, d))
    scale = sum((a.abs() * b.double().abs()).sum().item() for a, b in zip(Jd, gr))
    assert abs(lhs - rhs) < 2e-4 * scale, (lhs, rhs, scale)


def same_to_an_ulp(a, b):
    """Equal except for elements one or two ulps of the largest value apart (two instantiations of the same source may
    contract FMAs differently): max difference <= 2 ulp of the largest value, rel-L2 <= 1e-7."""
    diff = (a - b).abs()
    ulp = float(b.abs().max()) * 2.0 ** -23
    assert float(diff.max()) <= 2 * ulp and float(diff.double().norm() / b.double().norm()) <= 1e-7


@pytest.mark.parametrize('nx,ny,batch', [(64, 1024, 3), (256, 1024, 2), (1024, 1024, 4), (128, 64, 5), (64, 128, 3), (512, 256, 2), (256, 512, 2)])
def test_fused_both_residuals(nx, ny, batch, gpu_device):
    """nns_residual_both_f32 (spectral column pass + ONE row pass that also evaluates the 5-point stencil; every row length:
    one row per wave at ny = 1024, 2 ... 16 rows per wave below, incl. a last iteration with idle lines):
    the spectral outputs equal nns_spec_residual_f32 (same source, another instantiation: isolated elements may differ by
    an ulp of the result), the stencil outputs equal nns_fd_residual_f32 to rounding and the oracle to 1e-5 -- non-square grids, several grids per batch (the i-1 / i+1 rows wrap inside each
    grid), the split form (row pass alone), the all-float32 mode, and the engine's dispatch."""
    from nns import ops
    from nns.periodic import ResidualEngine
    from oracle import periodic as OP
    rng = np.random.default_rng(nx + ny)
    x = 2 * np.pi * np.arange(nx)[:, None] / nx
    y = 2 * np.pi * np.arange(ny)[None, :] / ny
    f = []
    for q in range(5):
        a = np.stack([np.cos((1 + b) * x + q) * np.sin(2 * y - b) + 0.05 * rng.standard_normal((nx, ny)) for b in range(batch)])
        f.append(a.astype(np.float32))
    f[3] = (f[0] - 1e-3 * f[3]).astype(np.float32); f[4] = (f[1] - 1e-3 * f[4]).astype(np.float32)     # u_prev, v_prev close to u, v
    d = [dev(a) for a in f]
    Lx, Ly = 2 * np.pi * nx / ny, 2 * np.pi
    fd_ref = ops.fd_residual(*d, DT, Lx / nx, Ly / ny, RHO, NU, 5)
    for precise in (2, True, False):                                         # float64 forward transforms, the library's pick, all-float32
        sp_ref = ops.spec_residual(*d, DT, Lx, Ly, RHO, NU, precise)
        fo, so = ops.residual_both(*d, DT, Lx, Ly, RHO, NU, precise)
        for a, b in zip(so, sp_ref):
            same_to_an_ulp(a, b)
        for a, b in zip(fo, fd_ref):
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
    so2 = ops.spec_residual_xpass(d[0], d[1], d[2], Lx, RHO, NU)
    fo2, so2 = ops.residual_both(*d, DT, Lx, Ly, RHO, NU, out_spec=so2, rowpass_only=True)
    assert all(torch.equal(a, b) for a, b in zip(fo2, fo)) or all(rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-7 for a, b in zip(fo2, fo))
    for a, b in zip(so2, ops.spec_residual(*d, DT, Lx, Ly, RHO, NU)):
        same_to_an_ulp(a, b)
    want = OP.fd_residual(*[a[0].astype(np.float64) for a in f], DT, Lx / nx, Ly / ny, RHO, NU, 5)
    for a, b in zip(ops.residual_both(*d, DT, Lx, Ly, RHO, NU)[0], want):
        assert rel_l2(a[0].cpu().numpy(), b) < 1e-5
    eng = ResidualEngine(nx, ny, DT, RHO, NU, Lx, Ly)
    (e_fd, e_sp), (s_fd, s_sp) = eng.both(*d), eng.both(*d, fused=False)
    for a, b in zip(e_sp, s_sp):
        same_to_an_ulp(a, b)
    assert all(rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6 for a, b in zip(e_fd, s_fd))
    # a row length the FFT engine does not serve: the same call IS the two separate calls (stencil kernel + spectral passes, dense along y)
    d48 = [t[:, :, :48].contiguous() for t in d]
    f48, s48 = ops.residual_both(*d48, DT, Lx, Ly, RHO, NU)
    for a, b in zip(list(f48) + list(s48), list(ops.fd_residual(*d48, DT, Lx / nx, Ly / 48, RHO, NU, 5)) + list(ops.spec_residual(*d48, DT, Lx, Ly, RHO, NU))):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):                                          # beyond both engines: loud
        z = torch.zeros(1, 8, 4100, device='cuda')
        ops.residual_both(z, z, z, z, z, DT, Lx, Ly, RHO, NU)


@pytest.mark.parametrize('nx,ny', [(64, 52), (128, 20), (1024, 12), (256, 8)])
def test_column_pass_ragged_column_counts(nx, ny, gpu_device):
    """The column pass needs nx to be a power of two but takes ANY number of columns (a column slab of a sharded grid, nns.slab):
    tiles of 8 * (1024 / nx) columns with a ragged last tile (masked stores, clamped loads), fewer columns than one tile, and the
    role-split kernel's memory waves on a partly filled tile -- against oracle.periodic.spectral_xpart."""
    from nns import ops
    from oracle import periodic as OP
    rng = np.random.default_rng(nx * 1000 + ny)
    f = [rng.standard_normal((3, nx, ny)).astype(np.float32) for _ in range(3)]
    Lx, rho, nu = 2.5, 1.3, 0.05
    got = ops.spec_residual_xpass(*[dev(a) for a in f], Lx, rho, nu)
    ref = OP.spectral_xpart(*[a.astype(np.float64) for a in f], Lx, rho, nu)
    for g, r in zip(got, ref):
        assert g.shape == (3, nx, ny) and rel_l2(g.cpu().numpy(), r) <= TOL


def _rough_fields(B, nx, ny, seed):
    """Fields with energy at every wavenumber, generated on the device (batches too large to build on the host)."""
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    x = torch.arange(nx, device='cuda', dtype=torch.float32)[:, None] * (2 * np.pi / nx)
    y = torch.arange(ny, device='cuda', dtype=torch.float32)[None, :] * (2 * np.pi / ny)
    base = [torch.cos(x) * torch.sin(y), -torch.sin(x) * torch.cos(y), -0.25 * (torch.cos(2 * x) + torch.cos(2 * y))]
    f = [(b[None] + 0.05 * torch.randn(B, nx, ny, device='cuda', generator=g)).contiguous() for b in base]
    return f + [(f[0] * 0.999 + 0.001).contiguous(), (f[1] * 0.999 - 0.001).contiguous()]


@pytest.mark.parametrize('B,nx,ny', [(40, 512, 1024), (64, 512, 512), (256, 256, 256), (1024, 128, 128), (4096, 64, 64)])
def test_marching_row_pass_equals_separate_kernels(B, nx, ny, gpu_device):
    """Batches large enough for the MARCHING fused row pass (spec_rowmarch_kernel: every line walks a chunk of consecutive rows, the
    stencil's row above parked in LDS, the row below = the next row's prefetch; all-float32 mode), every row length: its spectral
    outputs equal the separate column + row pass, its stencil outputs the standalone stencil kernel, grid b of the batch equals
    the same grid evaluated in a small batch (one-row chunks) bit for bit, run-to-run bitwise; the oracle on one grid."""
    from nns import ops
    from oracle import periodic as OP
    d = _rough_fields(B, nx, ny, seed=B + nx)
    Lx, Ly = 2 * np.pi * nx / ny, 2 * np.pi
    fo, so = ops.residual_both(*d, DT, Lx, Ly, RHO, NU, precise=False)
    fo2, so2 = ops.residual_both(*d, DT, Lx, Ly, RHO, NU, precise=False)
    sp = ops.spec_residual(*d, DT, Lx, Ly, RHO, NU, precise=False)
    fd = ops.fd_residual(*d, DT, Lx / nx, Ly / ny, RHO, NU, 5)
    small = [t[B - 2:].contiguous() for t in d]
    fs, ss = ops.residual_both(*small, DT, Lx, Ly, RHO, NU, precise=False)                     # 2 grids: the same kernel with one-row chunks
    for k in range(3):
        assert torch.equal(fo[k], fo2[k]) and torch.equal(so[k], so2[k])
        same_to_an_ulp(so[k], sp[k])
        assert torch.equal(so[k][B - 2:], ss[k]) and torch.equal(fo[k][B - 2:], fs[k])       # chunk length does not change a bit
        assert rel_l2(fo[k].cpu().numpy(), fd[k].cpu().numpy()) < 1e-6
    f64 = [t[B - 1].cpu().numpy().astype(np.float64) for t in d]
    ref_fd, ref_sp = OP.fd_residual(*f64, DT, Lx / nx, Ly / ny, RHO, NU, 5), OP.spectral_residual(*f64, DT, Lx, Ly, RHO, NU)
    for k in range(3):
        assert rel_l2(fo[k][B - 1].cpu().numpy(), ref_fd[k]) <= TOL and rel_l2(so[k][B - 1].cpu().numpy(), ref_sp[k]) <= TOL


@pytest.mark.parametrize('precise', [0, 2])
def test_marching_row_pass_on_a_row_slab(precise, gpu_device):
    """The marching row pass on a row slab (nx_local = 44 rows: chunks of 8 with a ragged last one; rows above / below the slab from the
    halo messages) against the same rows of the full-grid evaluation.  precise = 2: the float64-forward instantiation of the halo row
    pass (spec_ypass_kernel<N, double, true> with the halo rows), which the slab path takes when the viscous amplification exceeds 8."""
    from nns import ops
    B, nx, ny, nl, r0 = 400, 1024, 1024, 44, 100
    d = _rough_fields(1, nx, ny, seed=3)
    full_fd, full_sp = ops.residual_both(*d, DT, L, L, RHO, NU, precise=precise)
    part = ops.spec_residual_xpass(d[0], d[1], d[2], L, RHO, NU, precise=precise)
    loc = [t[:, r0:r0 + nl].expand(B, nl, ny).contiguous() for t in d]
    pl = [t[:, r0:r0 + nl].expand(B, nl, ny).contiguous() for t in part]
    top = torch.stack([t[:, r0 - 1].expand(B, ny) for t in d[:3]]).contiguous()
    bot = torch.stack([t[:, r0 + nl].expand(B, ny) for t in d[:3]]).contiguous()
    hf, hs = ops.residual_both_rowpass_halo(*loc, top, bot, pl, DT, L / nx, L, RHO, NU, precise=precise)
    for a, b in zip(hf, full_fd):
        assert bool((a == a[0:1]).all())                                                       # every grid of the batch the same
        assert rel_l2(a[B - 1].cpu().numpy(), b[0, r0:r0 + nl].cpu().numpy()) < 1e-6
    for a, b in zip(hs, full_sp):
        assert bool((a == a[0:1]).all())
        same_to_an_ulp(a[B - 1], b[0, r0:r0 + nl])


@pytest.mark.parametrize('n', [256, 1024])
@pytest.mark.parametrize('nu', [2 * np.pi / 1000, 0.1, 1.0])
def test_all_float32_mode_accuracy_and_the_precise_policy(n, nu, gpu_device):
    """The all-float32 spectral mode (transforms of forward-differenced lines, bounded filters) against the float64 oracle on smooth and
    on rough fields: within the 1e-5 bar at every viscosity.  precise=1 is the library's pick: bitwise the all-float32 result while
    nu pi N / (sqrt(3) L) <= 8, bitwise the float64-forward result (precise=2) above."""
    from nns import ops
    from oracle import periodic as OP
    rng = np.random.default_rng(n)
    for rough in (False, True):
        f = inputs(2, n)
        if rough:
            f = [a + (0.02 * rng.standard_normal(a.shape)).astype(np.float32) for a in f]
        d = [dev(a) for a in f]
        ref = OP.spectral_residual(*[a.astype(np.float64) for a in f], DT, L, L, 1.3, nu)
        r0, r1, r2 = (ops.spec_residual(*d, DT, L, L, 1.3, nu, precise=pr) for pr in (0, 1, 2))
        for g, r in zip(r0, ref):
            assert rel_l2(g.cpu().numpy(), r) <= TOL
        for g, r in zip(r2, ref):
            assert rel_l2(g.cpu().numpy(), r) <= 1e-6
        pick = r0 if nu * np.pi * n / (np.sqrt(3) * L) <= 8 else r2
        assert all(torch.equal(a, b) for a, b in zip(r1, pick))


@pytest.mark.parametrize('B', [3, 300, 1500])
def test_marching_row_pass_chunk_lengths(B, gpu_device):
    """The chunk length of the marching row pass follows the batch (64 x 256 grids: one-row chunks at B = 3, two rows at 300, eight at
    1500): the same answers as the separate kernels at each, white-noise fields."""
    from nns import ops
    g = torch.Generator(device='cuda'); g.manual_seed(B)
    nx, ny = 64, 256
    d = [torch.randn(B, nx, ny, device='cuda', generator=g) for _ in range(3)]
    d += [d[0] * 0.999 + 0.001, d[1] * 0.999 - 0.001]
    Lx, Ly = 2 * np.pi * nx / ny, 2 * np.pi
    fo, so = ops.residual_both(*d, DT, Lx, Ly, 1.3, NU, precise=False)
    sp = ops.spec_residual(*d, DT, Lx, Ly, 1.3, NU, precise=False)
    fd = ops.fd_residual(*d, DT, Lx / nx, Ly / ny, 1.3, NU, 5)
    for k in range(3):
        same_to_an_ulp(so[k], sp[k])
        assert rel_l2(fo[k].cpu().numpy(), fd[k].cpu().numpy()) < 1e-6


@pytest.mark.parametrize('precise', [0, 2])
def test_column_pass_segments_beyond_32bit_offsets(precise, gpu_device):
    """The role-split column pass addresses with 32-bit lane offsets inside a grid; a segmented (slab) layout whose source-rank
    blocks lie >= 2^32 elements apart (the library then takes the kernel with 64-bit row offsets) must give the same numbers as the
    plain column pass.  Also the float64-forward instantiation of the segmented layout (precise = 2)."""
    import ctypes
    from nns import ops, _lib
    B, nx, ny, seg = 2, 64, 64, 32
    stride = (1 << 32) + 4096                                 # elements between the two source-rank blocks of one field
    per = B * seg * ny                                        # one field of one source rank
    need = stride + 3 * per + 64
    free, _ = torch.cuda.mem_get_info()
    if free < 2 * need * 4 + (1 << 30):
        pytest.skip("needs 2 x %.1f GB of device memory" % (need * 4 / 1e9))
    f = [dev(a) for a in inputs(B, nx)[:3]]
    ref = ops.spec_residual_xpass(*f, L, RHO, NU, precise=precise)
    big_in = torch.empty(need, dtype=torch.float32, device='cuda')
    big_out = torch.empty(need, dtype=torch.float32, device='cuda')
    for k, t in enumerate(f):                                 # [src][field][grid][seg][ny] with the sources `stride` apart
        for src in range(nx // seg):
            big_in[src * stride + k * per: src * stride + (k + 1) * per].view(B, seg, ny).copy_(t[:, src * seg:(src + 1) * seg])
    q = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k * per)
    _lib.check(_lib.lib().nns_spec_residual_xpass_seg_f32(q(big_in, 0), q(big_in, 1), q(big_in, 2), q(big_out, 0), q(big_out, 1), q(big_out, 2),
                                                          B, nx, ny, seg, stride, L, RHO, NU, int(precise), torch.cuda.current_stream().cuda_stream),
               'nns_spec_residual_xpass_seg_f32')
    for k in range(3):
        for src in range(nx // seg):
            got = big_out[src * stride + k * per: src * stride + (k + 1) * per].view(B, seg, ny)
            assert torch.equal(got, ref[k][:, src * seg:(src + 1) * seg]), (k, src)


@pytest.mark.parametrize('B,nloc,ny,P,dtype', [(3, 5, 64, 4, torch.float32), (2, 7, 24, 4, torch.float32), (1, 4, 30, 3, torch.float64), (2, 3, 16, 2, torch.float64),
                                                 (4, 16, 1024, 8, torch.float32)])
def test_slab_transpose_pack_unpack_kernels(B, nloc, ny, P, dtype, gpu_device):
    """The all-to-all buffer layout [P][F][B][nloc][ny / P] against a torch restatement, both the 16-byte vector kernel (ny / P a multiple of the
    vector width) and the element kernel (ny / P = 6: not), float32 and float64, and the round trip."""
    from nns import ops
    g = torch.Generator(device='cuda'); g.manual_seed(B * ny + P)
    fields = [torch.randn(B, nloc, ny, device='cuda', dtype=dtype, generator=g) for _ in range(3)]
    nyl = ny // P
    send = torch.empty(P, 3, B, nloc, nyl, device='cuda', dtype=dtype)
    ops.slab_transpose_pack(fields, send, P)
    want = torch.stack([torch.stack([f[:, :, d * nyl:(d + 1) * nyl] for f in fields]) for d in range(P)])
    assert torch.equal(send, want)
    back = [torch.zeros_like(f) for f in fields]
    ops.slab_transpose_unpack(send, back, P)
    for a, b in zip(back, fields):
        assert torch.equal(a, b)


def _to_seg(parts, P):
    """Row-slab partials [B, nloc, ny] x 3 -> the return all-to-all's receive layout [P][3][B][nloc][ny / P]."""
    nyl = parts[0].shape[2] // P
    return torch.stack([torch.stack([t[:, :, s * nyl:(s + 1) * nyl] for t in parts]) for s in range(P)]).contiguous()


@pytest.mark.parametrize('ny,P,B', [(1024, 8, 5), (1024, 8, 800), (1024, 2, 5), (1024, 16, 5), (256, 4, 5), (256, 4, 3000), (64, 16, 5), (128, 1, 5)])
@pytest.mark.parametrize('precise', [0, 2])
def test_row_passes_read_segmented_partials_in_place(ny, P, B, precise, gpu_device):
    """Round 4: the slab step's row passes read the column-pass partials where the return all-to-all delivers them -- [src][field][grid][row]
    [ny / P] -- instead of from a copy scattered into row slabs.  Fused (marching / float64-forward) and plain spectral row pass, every
    piece length the layout admits (ny / P from the whole row down to one lane group, ny / 16), one-row chunks (B = 5) and marching chunks of
    several rows with a ragged last one (B = 800 / 3000 x nloc = 11), a halo message shared by a larger batch: BITWISE the in-place forms."""
    from nns import ops
    nloc = 11
    d = _rough_fields(B, nloc, ny, seed=ny + P)
    g = torch.Generator(device='cuda'); g.manual_seed(7)
    parts = [torch.randn(B, nloc, ny, device='cuda', generator=g) for _ in range(3)]
    Bh, g0 = B + 3, 2                                                        # the halo messages cover a larger batch: this call is grids 2 .. 6 of it
    top = torch.randn(3, Bh, ny, device='cuda', generator=g)
    bot = torch.randn(3, Bh, ny, device='cuda', generator=g)
    got = _to_seg(parts, P)
    ref_parts = [t.clone() for t in parts]
    ref_fd, ref_sp = ops.residual_both_rowpass_halo(*d, top, bot, ref_parts, DT, L / 1024, L, RHO, NU, precise=precise, halo_grid0=g0)
    fd, sp = ops.residual_both_rowpass_halo_seg(*d, top, bot, got, DT, L / 1024, L, RHO, NU, precise=precise, halo_grid0=g0)
    for a, b in zip(list(fd) + list(sp), list(ref_fd) + list(ref_sp)):
        assert torch.equal(a, b)
    ref_y = ops.spec_residual_ypass_(*d, *[t.clone() for t in parts], DT, L, RHO, NU, precise=precise)
    y = ops.spec_residual_ypass_seg(*d, got, DT, L, RHO, NU, precise=precise)
    for a, b in zip(y, ref_y):
        assert torch.equal(a, b)
    assert torch.equal(got, _to_seg(parts, P))                                # the receive buffer is read only


def test_segmented_row_pass_refuses_what_it_cannot_address(gpu_device):
    from nns import ops, _lib
    d = _rough_fields(2, 8, 256, seed=1)
    top = bot = torch.zeros(3, 2, 256, device='cuda')
    with pytest.raises(_lib.NnsError, match='seg_cols'):                       # 32 ranks: pieces of 8 columns < one lane group of 16
        ops.residual_both_rowpass_halo_seg(*d, top, bot, torch.zeros(32, 3, 2, 8, 8, device='cuda'), DT, 0.1, L, RHO, NU)
    d96 = _rough_fields(2, 8, 96, seed=1)
    with pytest.raises(_lib.NnsError, match='power of two'):
        ops.spec_residual_ypass_seg(*d96, torch.zeros(2, 3, 2, 8, 48, device='cuda'), DT, L, RHO, NU)


@pytest.mark.parametrize('B,nloc,ny,P,dtype,g0,Bc', [(5, 6, 64, 4, torch.float32, 1, 3), (2, 3, 16, 2, torch.float64, 0, 2), (64, 16, 1024, 8, torch.float32, 32, 32),
                                                       (3, 4, 32, 8, torch.float32, 0, 1)])
def test_slab_pack_halo_one_launch(B, nloc, ny, P, dtype, g0, Bc, gpu_device):
    """The fused pack + halo launch (nns_slab_pack_halo_*): a batch chunk into the all-to-all send buffer and the edge rows of the WHOLE batch
    into the two halo messages, against the separate pack / gather kernels; without halo buffers only the pack; HipCompute.pack_halo's
    fallback (ny / P not a vector multiple) gives the same."""
    from nns import ops
    from nns.slab import HipCompute
    g = torch.Generator(device='cuda'); g.manual_seed(B + ny)
    fields = [torch.randn(B, nloc, ny, device='cuda', dtype=dtype, generator=g) for _ in range(3)]
    nyl = ny // P
    want = torch.empty(P, 3, Bc, nloc, nyl, device='cuda', dtype=dtype)
    ops.slab_transpose_pack([t[g0:g0 + Bc].contiguous() for t in fields], want, P)
    wf = torch.stack([t[:, 0] for t in fields]).contiguous(); wl = torch.stack([t[:, -1] for t in fields]).contiguous()
    for use_compute in (False, True):
        send = torch.zeros_like(want); first = torch.zeros(3, B, ny, device='cuda', dtype=dtype); last = torch.zeros_like(first)
        (HipCompute().pack_halo if use_compute else ops.slab_pack_halo)(fields, send, first, last, g0, P)
        assert torch.equal(send, want) and torch.equal(first, wf) and torch.equal(last, wl)
        send.zero_()
        (HipCompute().pack_halo if use_compute else ops.slab_pack_halo)(fields, send, None, None, g0, P)
        assert torch.equal(send, want)
    odd = [t[:, :, :24].contiguous() for t in fields] if ny >= 32 else None
    if odd is not None and dtype == torch.float32:                              # ny / P = 6: the separate kernels behind the same call
        send = torch.zeros(4, 3, Bc, nloc, 6, device='cuda', dtype=dtype); first = torch.zeros(3, B, 24, device='cuda', dtype=dtype); last = torch.zeros_like(first)
        HipCompute().pack_halo(odd, send, first, last, g0, 4)
        w2 = torch.empty_like(send)
        ops.slab_transpose_pack([t[g0:g0 + Bc].contiguous() for t in odd], w2, 4)
        assert torch.equal(send, w2) and torch.equal(first, torch.stack([t[:, 0] for t in odd])) and torch.equal(last, torch.stack([t[:, -1] for t in odd]))


def test_dense_path_table_is_built_outside_a_stream_capture(gpu_device):
    """the circulant table of an axis length the FFT engine does not serve is built (host loop, allocation, synchronous copy) at the
    first call that needs it -- impossible inside a HIP graph capture.  A capture that meets a missing table is refused with
    NNS_ERR_UNSUPPORTED (and stays intact); after nns_spec_dense_warmup the same step captures, and the replayed graph gives the eager result."""
    from nns import ops, _lib
    n, L2 = 57, 3.21                                         # a size / box length no other test uses: its table cannot exist yet
    g = torch.Generator(device='cuda'); g.manual_seed(57)
    d = [torch.randn(2, n, n, device='cuda', generator=g) for _ in range(5)]
    out = tuple(torch.empty_like(d[0]) for _ in range(3))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with pytest.raises(_lib.NnsError, match='stream capture'):
            with torch.cuda.graph(graph, stream=side):
                ops.spec_residual(*d, DT, L2, L2, RHO, NU, out=out)
    torch.cuda.synchronize()
    assert _lib.lib().nns_spec_dense_warmup(n, L2) == 0
    ref = [t.clone() for t in ops.spec_residual(*d, DT, L2, L2, RHO, NU)]
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ops.spec_residual(*d, DT, L2, L2, RHO, NU, out=out)
    for t in out:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
