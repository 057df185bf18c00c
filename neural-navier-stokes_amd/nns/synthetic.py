"""Synthetic inputs for the periodic-box residual: decaying Taylor-Green
vortex on [0, 2 pi)^2 at time t (and t - dt for the previous step) plus band-limited noise
(|k| <= kmax, peak amplitude ``noise``), one seed per grid (seed0 + b).  Host-side NumPy: input
generation is not part of the hot path."""
import numpy as np


def taylor_green(n, t, nu, rho=1.0):
    x = 2 * np.pi * np.arange(n) / n
    X, Y = np.meshgrid(x, x, indexing='ij')
    F = np.exp(-2 * nu * t)
    return (np.cos(X) * np.sin(Y) * F, -np.sin(X) * np.cos(Y) * F,
            -rho / 4. * (np.cos(2 * X) + np.cos(2 * Y)) * F * F)


def band_limited_noise(n, rng, kmax=32, amp=0.05):
    k = np.fft.fftfreq(n, 1. / n)
    kr = np.fft.rfftfreq(n, 1. / n)
    mask = (k[:, None] ** 2 + kr[None, :] ** 2) <= kmax ** 2
    c = (rng.standard_normal((n, n // 2 + 1)) + 1j * rng.standard_normal((n, n // 2 + 1))) * mask
    f = np.fft.irfft2(c, s=(n, n))
    return amp * f / np.abs(f).max()


def residual_inputs(batch, n, t=0.1, dt=1e-3, nu=2 * np.pi / 1000, rho=1.0, seed0=1234, noise=0.05, kmax=32,
                    dtype=np.float32):
    """Returns u, v, p, u_prev, v_prev as [batch, n, n] arrays of ``dtype``."""
    u0, v0, p0 = taylor_green(n, t, nu, rho)
    up0, vp0, _ = taylor_green(n, t - dt, nu, rho)
    out = [np.empty((batch, n, n), dtype=dtype) for _ in range(5)]
    for b in range(batch):
        rng = np.random.default_rng(seed0 + b)
        nu_, nv_, np_ = (band_limited_noise(n, rng, min(kmax, n // 2 - 1), noise) for _ in range(3))
        out[0][b], out[1][b], out[2][b] = u0 + nu_, v0 + nv_, p0 + np_
        out[3][b], out[4][b] = up0 + nu_, vp0 + nv_
    return tuple(out)
