"""CPU pins of the oracle's CORRECTED Chebyshev path (oracle/chorin_spectral.py, Setup(corrected=True); an option of the build,
SURVEY.md section 8 (f) rank 3 -- no reference counterpart, so the pins are analytic): Neumann and inhomogeneous boundary data
folded into every derivative, and the well-conditioned projection step."""
import numpy as np
import pytest

from oracle import chorin_spectral as OS

SIDES = ('left', 'right', 'top', 'bottom')


def setup(N, kinds, values):
    h = 2. / N
    t = [(k, s, v, h, h) for k, v, s in zip(kinds, values, SIDES)]
    return OS.Setup(N, N, t, t, corrected=True)


@pytest.mark.parametrize('N', [17, 33])
def test_predictor_analytic_answers(N):
    dt, eps = 1e-3, 1e-7
    fac = (2 - 2 * np.pi**2 * dt) / (2 + 2 * np.pi**2 * dt)
    S = setup(N, ['neumann'] * 4, [0.0] * 4)
    x, y = S.x_i[:, None], S.y_i[None, :]
    f = np.cos(np.pi * x) * np.cos(np.pi * y)
    ui, _ = OS.predictor_step(S, eps * f, eps * f, eps * f, eps * f, dt)
    assert np.abs((ui / eps - fac * f)[1:-1, 1:-1]).max() < 1e-8 and np.abs((ui / eps - fac * f)[0, 1:-1]).max() < 1e-8
    S = setup(N, ['dirichlet', 'dirichlet', 'neumann', 'neumann'], [0.0, eps, 0.0, 0.0])
    f = eps * (1 + x) / 2 * np.ones_like(y)
    ui, _ = OS.predictor_step(S, f, f, f, f, dt)
    assert np.abs(ui - f)[:, 1:-1].max() < 1e-7 * eps
    S = setup(N, ['neumann'] * 4, [eps, eps, 0.0, 0.0])
    f = eps * x * np.ones_like(y)
    ui, _ = OS.predictor_step(S, f, f, f, f, dt)
    assert np.abs(ui - f)[:, 1:-1].max() < 1e-7 * eps
    with pytest.raises(NotImplementedError):                     # the reference path refuses Neumann data (:218-221)
        OS.Setup(N, N, [('neumann', s, 0.0, 0.1, 0.1) for s in SIDES], [('dirichlet', s, 0.0, 0.1, 0.1) for s in SIDES])


def test_homogeneous_dirichlet_corrected_path_unchanged_by_the_folding():
    """With homogeneous Dirichlet data the folded operators are the interior blocks and every constant is zero."""
    N = 17
    S = setup(N, ['dirichlet'] * 4, [0.0] * 4)
    o = S.folded['u']
    assert np.array_equal(o['Dx'], S.Dx[1:-1, 1:-1]) and not np.any(o['cx']) and not np.any(o['cxx'])
    assert np.array_equal(S.helm['u']['Mx'], S.Dx_sqr[1:-1, 1:-1])


@pytest.mark.parametrize('N', [17, 33])
def test_projection_analytic_and_divergence(N):
    dt, rho = 1e-3, 1.3
    S = setup(N, ['dirichlet'] * 4, [0.0] * 4)
    xi = S.x_i[1:-1]
    for k in range(6):                                            # the interior pressure derivative is exact on polynomials
        assert np.abs(S.DPx @ xi**k - (k * xi**(k - 1) if k else 0 * xi)).max() < 1e-10
    lam = np.sort(np.abs(S.lpx))
    assert lam[0] < 1e-10 and abs(lam[1] - np.pi**2 / 4) < 1e-6
    x, y = S.x_i[:, None], S.y_i[None, :]
    phi = np.cos(np.pi * x) * np.cos(np.pi * y)
    ui = dt / rho * (-np.pi * np.sin(np.pi * x) * np.cos(np.pi * y))
    vi = dt / rho * (-np.pi * np.cos(np.pi * x) * np.sin(np.pi * y))
    u1, v1, p1 = OS.correction_step_corrected(S, ui, vi, np.zeros((N, N)), dt, rho)
    Q, ph = p1[1:-1, 1:-1], phi[1:-1, 1:-1]
    assert np.abs(u1[1:-1, 1:-1]).max() < 1e-8 * np.abs(ui).max()
    assert np.abs((Q - Q.mean()) - (ph - ph.mean())).max() < 1e-8
    rng = np.random.default_rng(1)
    ui, vi = rng.standard_normal((N, N)), rng.standard_normal((N, N))
    u1, v1, _ = OS.correction_step_corrected(S, ui, vi, np.zeros((N, N)), dt, rho)
    div0 = S.Dx[1:-1, :] @ ui[:, 1:-1] + vi[1:-1, :] @ S.Dy[1:-1, :].T
    div = S.Dx[1:-1, :] @ u1[:, 1:-1] + v1[1:-1, :] @ S.Dy[1:-1, :].T
    assert np.abs(div - div.mean()).max() < 1e-9 * np.abs(div0).max()
