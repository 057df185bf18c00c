"""Oracle (test infrastructure only): incompressible Navier-Stokes residual on a periodic box.

PARITY UNPINNED BY THE REFERENCE.  The reference (mhw32/neural-navier-stokes) has no periodic
Fourier path, no 9-point Laplacian and no assembled residual (SURVEY.md section 8, row a17;
motivation: src/neural_spectral/derivations/derivation.tex:25-59 "Neural Residual PDEs", NS
form src/direct_fd/derivations/derivation.tex:64-68).  These functions DEFINE the operators
the HIP residual engine must reproduce; they are pinned by analytic known-answers
(Taylor-Green vortex: the spectral residual vanishes to O(nu^2 dt), the FD residual converges
at 2nd order) in tests/test_oracle_periodic.py rather than by reference fixtures.

Definitions (fields [..., nx, ny], axis 0 = x, axis 1 = y, periodic, x_i = i*Lx/nx):

  r_u   = (u - u_prev)/dt + u u_x + v u_y + p_x/rho - nu lap(u)
  r_v   = (v - v_prev)/dt + u v_x + v v_y + p_y/rho - nu lap(v)
  r_div = u_x + v_y

FD back-end: 2nd-order central first derivatives; lap = 5-point, or the 9-point
"Mehrstellen" form  d_xx + d_yy + (dx^2+dy^2)/12 * d_xx d_yy  (for dx == dy this is the classic
[1 4 1; 4 -20 4; 1 4 1]/(6 h^2) stencil).
Spectral back-end: d/dx <-> i*kx (the Nyquist mode of an even-length axis is zeroed for odd
derivatives, the usual convention that keeps the result real), lap <-> -(kx^2 + ky^2).
"""
import numpy as np


def _r(f, s, axis):
    return np.roll(f, s, axis=axis)


def fd_derivs(f, dx, dy, stencil=5):
    """Returns (f_x, f_y, lap f) with periodic wrap."""
    xm, xp = _r(f, 1, -2), _r(f, -1, -2)
    ym, yp = _r(f, 1, -1), _r(f, -1, -1)
    fx = (xp - xm) / (2 * dx)
    fy = (yp - ym) / (2 * dy)
    dxx = (xp - 2 * f + xm)
    dyy = (yp - 2 * f + ym)
    lap = dxx / dx**2 + dyy / dy**2
    if stencil == 9:
        corners = (_r(xm, 1, -1) + _r(xm, -1, -1) + _r(xp, 1, -1) + _r(xp, -1, -1))
        dxxdyy = corners - 2 * (xm + xp + ym + yp) + 4 * f
        lap = lap + (dx**2 + dy**2) / 12. * dxxdyy / (dx**2 * dy**2)
    elif stencil != 5:
        raise ValueError("stencil must be 5 or 9")
    return fx, fy, lap


def fd_residual(u, v, p, u_prev, v_prev, dt, dx, dy, rho, nu, stencil=5):
    ux, uy, lu = fd_derivs(u, dx, dy, stencil)
    vx, vy, lv = fd_derivs(v, dx, dy, stencil)
    px, py, _ = fd_derivs(p, dx, dy, 5)
    r_u = (u - u_prev) / dt + u * ux + v * uy + px / rho - nu * lu
    r_v = (v - v_prev) / dt + u * vx + v * vy + py / rho - nu * lv
    return r_u, r_v, ux + vy


def wavenumbers(n, L):
    """(k for odd derivatives [Nyquist zeroed], k for even derivatives), length n, FFT order."""
    k = 2 * np.pi * np.fft.fftfreq(n, d=L / n)
    k1 = k.copy()
    if n % 2 == 0:
        k1[n // 2] = 0.0
    return k1, k


def spectral_derivs(f, Lx, Ly):
    """Returns (f_x, f_y, lap f) via rfft2 / irfft2 (computed in float64 complex)."""
    nx, ny = f.shape[-2], f.shape[-1]
    kx1, kx = wavenumbers(nx, Lx)
    ky1, ky = wavenumbers(ny, Ly)
    nyh = ny // 2 + 1
    F = np.fft.rfft2(f.astype(np.float64))
    KX1, KX = kx1[:, None], kx[:, None]
    KY1, KY = ky1[None, :nyh].copy(), np.abs(ky[None, :nyh])
    fx = np.fft.irfft2(1j * KX1 * F, s=(nx, ny))
    fy = np.fft.irfft2(1j * KY1 * F, s=(nx, ny))
    lap = np.fft.irfft2(-(KX**2 + KY**2) * F, s=(nx, ny))
    return fx, fy, lap


def spectral_residual(u, v, p, u_prev, v_prev, dt, Lx, Ly, rho, nu):
    ux, uy, lu = spectral_derivs(u, Lx, Ly)
    vx, vy, lv = spectral_derivs(v, Lx, Ly)
    px, py, _ = spectral_derivs(p, Lx, Ly)
    r_u = (u - u_prev) / dt + u * ux + v * uy + px / rho - nu * lu
    r_v = (v - v_prev) / dt + u * vx + v * vy + py / rho - nu * lv
    return r_u, r_v, ux + vy


def residual_vjp(u, v, g_u, g_v, g_div, dt, rho, nu, derivs):
    """Vector-Jacobian product of the residual (either back-end): given g_* = dLoss/dr_*, returns
    (dLoss/du, dLoss/dv, dLoss/dp, dLoss/du_prev, dLoss/dv_prev).  `derivs(f) -> (f_x, f_y, lap f)` is the
    back-end's linear operator triple.  Both back-ends have antisymmetric first derivatives (periodic central
    difference; i k with the Nyquist mode dropped) and a symmetric Laplacian, so D^T = -D, L^T = L:

      grad_u = g_u/dt + g_u u_x + g_v v_x - D_x(g_u u) - D_y(g_u v) - nu L g_u - D_x g_div
      grad_v = g_v/dt + g_u u_y + g_v v_y - D_x(g_v u) - D_y(g_v v) - nu L g_v - D_y g_div
      grad_p = -(D_x g_u + D_y g_v)/rho,   grad_u_prev = -g_u/dt,   grad_v_prev = -g_v/dt

    (p enters the residual linearly, so the product does not depend on it).  Pinned by the directional-derivative
    identity <J d, g> = <d, J^T g> in tests/test_oracle_periodic.py -- the residual is quadratic, so a central
    difference gives J d exactly."""
    ux, uy, _ = derivs(u)
    vx, vy, _ = derivs(v)
    aux, _, _ = derivs(g_u * u)
    _, avy, _ = derivs(g_u * v)
    bux, _, _ = derivs(g_v * u)
    _, bvy, _ = derivs(g_v * v)
    ax, _, la = derivs(g_u)
    _, by, lb = derivs(g_v)
    dx_, dy_, _ = derivs(g_div)
    grad_u = g_u / dt + g_u * ux + g_v * vx - aux - avy - nu * la - dx_
    grad_v = g_v / dt + g_u * uy + g_v * vy - bux - bvy - nu * lb - dy_
    grad_p = -(ax + by) / rho
    return grad_u, grad_v, grad_p, -g_u / dt, -g_v / dt


def fd_residual_vjp(u, v, g_u, g_v, g_div, dt, dx, dy, rho, nu, stencil=5):
    """VJP of fd_residual.  The pressure gradient always uses the 5-point first derivatives (as fd_residual does);
    first derivatives do not depend on `stencil`, so one operator triple serves all terms."""
    return residual_vjp(u, v, g_u, g_v, g_div, dt, rho, nu, lambda f: fd_derivs(f, dx, dy, stencil))


def spectral_residual_vjp(u, v, g_u, g_v, g_div, dt, Lx, Ly, rho, nu):
    """VJP of spectral_residual."""
    return residual_vjp(u, v, g_u, g_v, g_div, dt, rho, nu, lambda f: spectral_derivs(f, Lx, Ly))


def taylor_green(nx, ny, t, nu, rho=1.0, Lx=2 * np.pi, Ly=2 * np.pi):
    """Analytic decaying Taylor-Green vortex on [0,Lx)x[0,Ly) (exact NS solution for
    Lx = Ly = 2*pi): u = cos x sin y F, v = -sin x cos y F, p = -rho/4 (cos 2x + cos 2y) F^2,
    F = exp(-2 nu t)."""
    x = Lx * np.arange(nx) / nx
    y = Ly * np.arange(ny) / ny
    X, Y = np.meshgrid(x, y, indexing='ij')
    F = np.exp(-2 * nu * t)
    u = np.cos(X) * np.sin(Y) * F
    v = -np.sin(X) * np.cos(Y) * F
    p = -rho / 4. * (np.cos(2 * X) + np.cos(2 * Y)) * F * F
    return u, v, p


def spectral_xpart(u, v, p, Lx, rho, nu):
    """x-direction part of the spectral residual (what the HIP x-pass leaves in r_u, r_v, r_div):
    P_u = u u_x + p_x/rho - nu u_xx,  P_v = u v_x - nu v_xx,  P_d = u_x."""
    nx = u.shape[-2]
    k1, k = wavenumbers(nx, Lx)
    def d(f, mult):
        return np.fft.ifft(np.fft.fft(f.astype(np.float64), axis=-2) * mult[:, None], axis=-2).real
    ux, vx, px = d(u, 1j * k1), d(v, 1j * k1), d(p, 1j * k1)
    uxx, vxx = d(u, -(k * k)), d(v, -(k * k))
    return u * ux + px / rho - nu * uxx, u * vx - nu * vxx, ux


def spectral_ypart(u, v, p, u_prev, v_prev, pu, pv, pd, dt, Ly, rho, nu):
    """y-direction completion: r_u = (u-u_prev)/dt + P_u + v u_y - nu u_yy, etc."""
    ny = u.shape[-1]
    k1, k = wavenumbers(ny, Ly)
    def d(f, mult):
        return np.fft.ifft(np.fft.fft(f.astype(np.float64), axis=-1) * mult, axis=-1).real
    uy, vy, py = d(u, 1j * k1), d(v, 1j * k1), d(p, 1j * k1)
    uyy, vyy = d(u, -(k * k)), d(v, -(k * k))
    r_u = (u - u_prev) / dt + pu + v * uy - nu * uyy
    r_v = (v - v_prev) / dt + pv + v * vy + py / rho - nu * vyy
    return r_u, r_v, pd + vy


def pinn_head(out, state, target, backend, consts, lam=1.0, w_div=1.0):
    """Loss head of the physics-informed step (float64): out, state, target [B, 3, nx, ny] (target None: no data term);
    pred = state + out (channels u, v, p); data = mean (pred - target)^2; phys = mean r_u^2 + mean r_v^2 + w_div mean r_div^2 of the
    residual of pred against (state[:, 0], state[:, 1]); total = data + lam phys.  backend 'fd' (consts = dt, dx, dy, rho, nu, stencil)
    or 'spectral' (consts = dt, Lx, Ly, rho, nu).  Returns (total, data, phys, d total / d out).
    The reference states this objective and never implements it (src/neural_spectral/derivations/derivation.tex:25-34)."""
    pred = state + out
    u, v, p = pred[:, 0], pred[:, 1], pred[:, 2]
    res, vjp = (fd_residual, fd_residual_vjp) if backend == 'fd' else (spectral_residual, spectral_residual_vjp)
    r = res(u, v, p, state[:, 0], state[:, 1], *consts)
    n = u.size
    phys = (r[0] ** 2).mean() + (r[1] ** 2).mean() + w_div * (r[2] ** 2).mean()
    g = vjp(u, v, 2 * lam / n * r[0], 2 * lam / n * r[1], 2 * lam * w_div / n * r[2], *consts)
    grad = np.stack(g[:3], axis=1)
    data = 0.0
    if target is not None:
        data = ((pred - target) ** 2).mean()
        grad = grad + 2.0 * (pred - target) / pred.size
    return data + lam * phys, data, phys, grad
