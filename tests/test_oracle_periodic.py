"""Analytic pins for the periodic-box residual oracle (oracle/periodic.py).

The reference has no periodic/Fourier/9-point/assembled-residual code (SURVEY.md section 8 row
a17), so these operators are pinned by known answers instead of reference fixtures:
  * Taylor-Green is an exact NS solution: the spectral residual is O(nu^2 dt) (time
    discretisation only), the divergence is zero to rounding;
  * the FD residual converges at 2nd order (5-point and 9-point);
  * the 9-point Laplacian equals the classic [1 4 1; 4 -20 4; 1 4 1]/(6h^2) stencil for dx == dy;
  * spectral derivatives of single Fourier modes are exact, the Nyquist mode is dropped for odd
    derivatives and kept for the Laplacian.
"""
import numpy as np
import pytest

from oracle import periodic as OP

NU, RHO, DT, T0 = 2 * np.pi / 1000, 1.0, 1e-3, 0.1
L = 2 * np.pi


def tg_inputs(n):
    u, v, p = OP.taylor_green(n, n, T0, NU, RHO)
    up, vp, _ = OP.taylor_green(n, n, T0 - DT, NU, RHO)
    return u, v, p, up, vp


def test_spectral_residual_vanishes_on_taylor_green():
    u, v, p, up, vp = tg_inputs(64)
    ru, rv, rd = OP.spectral_residual(u, v, p, up, vp, DT, L, L, RHO, NU)
    # (u - u_prev)/dt = u_t + O(dt u_tt) with u_tt = 4 nu^2 u  ->  residual ~ 2 nu^2 dt |u|
    bound = 2.5 * NU**2 * DT
    assert np.abs(ru).max() < bound and np.abs(rv).max() < bound
    assert np.abs(rd).max() < 1e-13


@pytest.mark.parametrize('stencil', [5, 9])
def test_fd_residual_second_order(stencil):
    errs = []
    for n in (32, 64, 128):
        u, v, p, up, vp = tg_inputs(n)
        h = L / n
        ru, rv, rd = OP.fd_residual(u, v, p, up, vp, DT, h, h, RHO, NU, stencil)
        errs.append(max(np.abs(ru).max(), np.abs(rv).max()))
    assert 3.6 < errs[0] / errs[1] < 4.4 and 3.6 < errs[1] / errs[2] < 4.4


def test_nine_point_is_classic_stencil():
    rng = np.random.default_rng(0)
    f = rng.standard_normal((12, 10))
    h = 0.3
    _, _, lap9 = OP.fd_derivs(f, h, h, 9)
    r = lambda a, sx, sy: np.roll(np.roll(a, sx, 0), sy, 1)
    edge = r(f, 1, 0) + r(f, -1, 0) + r(f, 0, 1) + r(f, 0, -1)
    corner = r(f, 1, 1) + r(f, 1, -1) + r(f, -1, 1) + r(f, -1, -1)
    classic = (4 * edge + corner - 20 * f) / (6 * h * h)
    np.testing.assert_allclose(lap9, classic, rtol=1e-12, atol=1e-11)


def test_spectral_derivs_single_modes_and_nyquist():
    n = 16
    x = L * np.arange(n) / n
    X, Y = np.meshgrid(x, x, indexing='ij')
    f = np.sin(3 * X) * np.cos(2 * Y)
    fx, fy, lap = OP.spectral_derivs(f, L, L)
    np.testing.assert_allclose(fx, 3 * np.cos(3 * X) * np.cos(2 * Y), atol=1e-12)
    np.testing.assert_allclose(fy, -2 * np.sin(3 * X) * np.sin(2 * Y), atol=1e-12)
    np.testing.assert_allclose(lap, -13 * f, atol=1e-11)
    nyq = np.cos((n // 2) * X) + 0 * Y                   # the x-Nyquist mode
    fx, fy, lap = OP.spectral_derivs(nyq, L, L)
    assert np.abs(fx).max() < 1e-12 and np.abs(fy).max() < 1e-12
    np.testing.assert_allclose(lap, -(n // 2) ** 2 * nyq, atol=1e-10)


def test_batched_leading_axes_and_non_square():
    rng = np.random.default_rng(1)
    f = rng.standard_normal((3, 2, 8, 12))
    fx, fy, lap = OP.spectral_derivs(f, L, 2 * L)
    a, b, c = OP.spectral_derivs(f[1, 1], L, 2 * L)
    np.testing.assert_allclose(fx[1, 1], a, atol=1e-13)
    np.testing.assert_allclose(lap[1, 1], c, atol=1e-12)
    gx, gy, gl = OP.fd_derivs(f, 0.1, 0.2, 9)
    a, b, c = OP.fd_derivs(f[2, 0], 0.1, 0.2, 9)
    np.testing.assert_array_equal(gl[2, 0], c)


def test_residual_vjp_is_the_adjoint_of_the_jacobian():
    """<J d, g> = <d, J^T g> for both back-ends and both stencils.  The residual is quadratic in (u, v) and linear in
    the rest, so J d = (r(w + d) - r(w - d)) / 2 exactly (no step-size error)."""
    rng = np.random.default_rng(5)
    nx, ny = 16, 12
    Lx, Ly, dt, rho, nu = 2.0, 3.0, 0.01, 1.3, 0.05
    dx, dy = Lx / nx, Ly / ny
    w = [rng.standard_normal((2, nx, ny)) for _ in range(5)]          # u, v, p, u_prev, v_prev
    d = [rng.standard_normal((2, nx, ny)) for _ in range(5)]
    g = [rng.standard_normal((2, nx, ny)) for _ in range(3)]
    cases = [
        (lambda *a: OP.fd_residual(*a, dt, dx, dy, rho, nu, 5), lambda *a: OP.fd_residual_vjp(*a, dt, dx, dy, rho, nu, 5)),
        (lambda *a: OP.fd_residual(*a, dt, dx, dy, rho, nu, 9), lambda *a: OP.fd_residual_vjp(*a, dt, dx, dy, rho, nu, 9)),
        (lambda *a: OP.spectral_residual(*a, dt, Lx, Ly, rho, nu), lambda *a: OP.spectral_residual_vjp(*a, dt, Lx, Ly, rho, nu)),
    ]
    for res, vjp in cases:
        rp = res(*[a + b for a, b in zip(w, d)])
        rm = res(*[a - b for a, b in zip(w, d)])
        Jd = [(a - b) / 2 for a, b in zip(rp, rm)]
        lhs = sum(float((a * b).sum()) for a, b in zip(Jd, g))
        grads = vjp(w[0], w[1], *g)
        rhs = sum(float((a * b).sum()) for a, b in zip(grads, d))
        assert abs(lhs - rhs) <= 1e-10 * max(1.0, abs(lhs)), (lhs, rhs)


@pytest.mark.parametrize('n', [16, 64, 256])
def test_forward_difference_filters_reproduce_the_spectral_derivatives(n):
    """The identity behind the all-float32 device mode (csrc/spectral_kernels.hip, deriv_core DIFF32), in float64 where it is exact to
    rounding: with D = FFT(d), d_j = f_{j+1} - f_j (periodic) and theta = 2 pi k / n,
        i k FFT(f) = M1 D,     M1 = (k/2) (cot(theta/2) - i)          (0 at k = 0 and, for odd derivatives, at the Nyquist mode)
        -k^2 FFT(f) = F2 D,    F2 = (k/2) (k + i k cot(theta/2))      (Nyquist mode kept: cot(pi/2) = 0)
    so the x-derivative and second derivative of oracle/periodic.py's spectral_derivs come out of ONE transform of the differenced line."""
    rng = np.random.default_rng(n)
    L = 2.5
    f = rng.standard_normal((3, n, n))
    fx, _, _ = OP.spectral_derivs(f, L, L)
    ks = 2 * np.pi / L
    k = np.fft.fftfreq(n, 1.0 / n)                                     # integer wavenumbers
    ko = k.copy(); ko[n // 2] = 0.0
    ct = np.zeros(n)
    nz = (k != 0) & (np.abs(k) != n // 2)
    ct[nz] = np.abs(k[nz]) / np.tan(np.pi * np.abs(k[nz]) / n)        # the device's table: |k| cot(pi |k| / n), even in k
    D = np.fft.fft(np.roll(f, -1, axis=1) - f, axis=1)
    M1 = 0.5 * ks * (ct - 1j * ko)
    F2 = 0.5 * ks * ks * (k * k + 1j * k * ct)
    got_x = np.fft.ifft(D * M1[None, :, None], axis=1).real
    got_xx = np.fft.ifft(D * F2[None, :, None], axis=1).real
    want_xx = np.fft.ifft(np.fft.fft(f, axis=1) * (-(ks * k) ** 2)[None, :, None], axis=1).real
    np.testing.assert_allclose(got_x, fx, atol=1e-10 * np.abs(fx).max())
    np.testing.assert_allclose(got_xx, want_xx, atol=1e-10 * np.abs(want_xx).max())


@pytest.mark.parametrize('backend', ['fd', 'spectral'])
def test_pinn_head_gradient_is_the_derivative_of_its_total(backend):
    """oracle pinn_head (the loss head of the physics-informed step): its gradient against a central difference of its own total
    (the total is a quartic polynomial of `out`: the difference quotient is accurate to O(eps^2))."""
    rng = np.random.default_rng(5)
    B, n = 2, 12
    out, state, target = (rng.standard_normal((B, 3, n, n)) * s for s in (0.1, 1.0, 1.0))
    consts = (1e-2, 0.3, 0.25, 1.3, 0.05, 9) if backend == 'fd' else (1e-2, 2 * np.pi, 3.0, 1.3, 0.05)
    for tg, w_div in ((target, 2.0), (None, 1.0)):
        total, data, phys, grad = OP.pinn_head(out, state, tg, backend, consts, lam=0.3, w_div=w_div)
        assert abs(total - (data + 0.3 * phys)) <= 1e-12 * abs(total) and (tg is not None or data == 0.0)
        d = rng.standard_normal(out.shape)
        eps = 1e-5
        num = (OP.pinn_head(out + eps * d, state, tg, backend, consts, 0.3, w_div)[0] - OP.pinn_head(out - eps * d, state, tg, backend, consts, 0.3, w_div)[0]) / (2 * eps)
        assert abs(num - (grad * d).sum()) <= 1e-6 * abs(num)
