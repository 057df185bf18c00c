"""The CPU oracle vs golden vectors captured from the reference (oracle/capture.py).

float64 throughout.  Stencil operators must agree to ~1 ulp-level (1e-13 rel); anything that goes
through a LAPACK factorisation in the reference (ADI solves, eig/inv) to 1e-9 / 1e-8.
"""
import numpy as np
import pytest

from conftest import load_golden, unpack_bcs, rel_l2, KINDS, SIDES
from oracle import boundary as OB
from oracle import chorin_fd as OC
from oracle import direct_fd as OD
from oracle import chorin_spectral as OS


# ------------------------------------------------------------------ boundary (a1)
def test_boundary_single():
    g = load_golden('boundary.npz')
    val, dx, dy = float(g['single_value']), float(g['single_dx']), float(g['single_dy'])
    for kind in KINDS:
        for side in SIDES:
            A = g['A0'].copy()
            r = OB.apply_bc(A, kind, side, val, dx, dy)
            assert r is A
            np.testing.assert_array_equal(A, g['%s_%s' % (kind, side)])


@pytest.mark.parametrize('name', ['u', 'v', 'p', 'mixed'])
def test_boundary_lists_corner_order(name):
    g = load_golden('boundary.npz')
    A = g['S0'].copy()
    OB.apply_bc_list(A, unpack_bcs(g, 'list_%s_bc' % name))
    np.testing.assert_array_equal(A, g['list_' + name])


# ------------------------------------------------------------------ chorin_fd (a2-a6)
@pytest.mark.parametrize('n', [16, 64])
def test_chorin_fd_operators(n):
    g = load_golden('chorin_fd_ops_%d.npz' % n)
    dt, rho, nu, beta, dx, dy = g['params']
    u, v, u1, v1, p0 = g['u'], g['v'], g['u1'], g['v1'], g['p0']
    ui, vi = OC.explicit_predictor(u, v, u1, v1, dt, dx, dy, nu)
    np.testing.assert_array_equal(ui, g['pred_explicit_ui'])
    np.testing.assert_array_equal(vi, g['pred_explicit_vi'])
    ui, vi = OC.semi_implicit_predictor(u, v, u1, v1, dt, dx, dy, nu)
    assert rel_l2(ui, g['pred_semi_implicit_ui']) < 1e-12
    assert rel_l2(vi, g['pred_semi_implicit_vi']) < 1e-12
    a, b = OC.correction(u, v, p0, dt, dx, dy)
    np.testing.assert_array_equal(a, g['corr_u'])
    np.testing.assert_array_equal(b, g['corr_v'])


@pytest.mark.parametrize('n', [16, 64])
@pytest.mark.parametrize('nit', [3, 50, 400])
def test_chorin_fd_sor_bitwise_and_sweep_count(n, nit):
    g = load_golden('chorin_fd_ops_%d.npz' % n)
    dt, rho, nu, beta, dx, dy = g['params']
    p = g['press_p0_nit%d' % nit].copy()
    r, (sweeps, err) = OC.get_pressure(g['press_ui'], g['press_vi'], p, dt, dx, dy, rho, beta, nit,
                                       return_info=True)
    assert r is p
    np.testing.assert_array_equal(p, g['press_p_nit%d' % nit])        # wavefront == lexicographic
    assert sweeps == int(g['press_sweeps_nit%d' % nit])
    assert err == float(g['press_err_nit%d' % nit])
    assert sweeps <= nit - 1                                          # 'it' starts at 1 (:183)


def test_sor_wavefront_equals_lexicographic_loop():
    rng = np.random.default_rng(5)
    p = rng.standard_normal((9, 12))
    C = rng.standard_normal((9, 12))
    a = OC.sor_sweep_wavefront(p.copy(), C, 0.2, 0.3, 1.25)
    b = OC.sor_sweep_lexicographic(p.copy(), C, 0.2, 0.3, 1.25)
    np.testing.assert_array_equal(a, b)


def test_sor_tolerance_path_is_exercised():
    g = load_golden('chorin_fd_ops_16.npz')
    assert int(g['press_sweeps_nit400']) < 399       # stopped by err <= tol, not by the cap
    assert float(g['press_err_nit400']) <= OC.SOR_TOL


@pytest.mark.parametrize('n', [16, 64])
@pytest.mark.parametrize('method', ['explicit', 'semi_implicit'])
def test_chorin_fd_full_step(n, method):
    g = load_golden('chorin_fd_ops_%d.npz' % n)
    dt, rho, nu, beta, dx, dy = g['params']
    u_bc, v_bc, p_bc = (unpack_bcs(g, k + '_bc') for k in 'uvp')
    p = 0.01 * g['p0'].copy()
    a, b, c, (sweeps, _) = OC.step(0.1 * g['u'], 0.1 * g['v'], 0.1 * g['u1'], 0.1 * g['v1'], p, u_bc, v_bc,
                                    p_bc, dt, dx, dy, rho, nu, beta, 20, method, return_info=True)
    tol = 0 if method == 'explicit' else 1e-11
    for got, key in ((a, 'u'), (b, 'v'), (c, 'p')):
        assert rel_l2(got, g['step_%s_%s' % (method, key)]) <= tol
    assert sweeps == int(g['step_%s_sweeps' % method])


@pytest.mark.parametrize('case', [(16, 'explicit', 0.1), (16, 'semi_implicit', 0.1), (64, 'explicit', 0.02),
                                  (64, 'explicit', 0.01), (64, 'semi_implicit', 0.02),
                                  (64, 'semi_implicit', 0.01)])
def test_chorin_fd_cavity_trajectory(case):
    n, method, nu = case
    g = load_golden('chorin_fd_cavity_%d_%s_nu%g.npz' % (n, method, nu))
    dt, rho, nu_, beta, dx, dy = g['params']
    u_bc, v_bc, p_bc = OB.cavity_bcs(dx, dy)
    z = np.zeros((n, n))
    ul, vl, pl = OC.simulate(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, int(g['nt']), int(g['nit']),
                             dt, rho, nu_, beta, method)
    sel = slice(None) if n == 16 else [0, -1]
    tol = 0 if method == 'explicit' else 1e-9
    assert rel_l2(ul[sel], g['u']) <= tol
    assert rel_l2(vl[sel], g['v']) <= tol
    assert rel_l2(pl[sel], g['p']) <= tol


# ------------------------------------------------------------------ direct_fd (a7-a9)
@pytest.mark.parametrize('n', [16, 64])
def test_direct_fd_operators(n):
    g = load_golden('direct_fd_ops_%d.npz' % n)
    dt, rho, nu, dx, dy = g['params']
    u_bc, v_bc, p_bc = (unpack_bcs(g, k + '_bc') for k in 'uvp')
    b = OD.build_up_b(g['u'], g['v'], dt, dx, dy, rho)
    np.testing.assert_array_equal(b, g['b'])
    for nit in (1, 50):
        p = g['p0'].copy()
        r = OD.pressure_poisson(p, 1e-3 * b, p_bc, dx, dy, nit)
        assert r is p
        np.testing.assert_array_equal(p, g['poisson_nit%d' % nit])
    u, v, p = 0.1 * g['u'], 0.1 * g['v'], 0.01 * g['p0']
    OD.step(u, v, p, u_bc, v_bc, p_bc, dt, dx, dy, rho, nu, 20)
    np.testing.assert_array_equal(u, g['step_u'])
    np.testing.assert_array_equal(v, g['step_v'])
    np.testing.assert_array_equal(p, g['step_p'])


@pytest.mark.parametrize('n', [16, 64])
def test_direct_fd_cavity_trajectory(n):
    g = load_golden('direct_fd_cavity_%d.npz' % n)
    dt, rho, nu, dx, dy = g['params']
    u_bc, v_bc, p_bc = OB.cavity_bcs(dx, dy)
    z = np.zeros((n, n))
    ul, vl, pl = OD.simulate(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, int(g['nt']), int(g['nit']),
                             dt, rho, nu)
    sel = slice(None) if n == 16 else [0, -1]
    np.testing.assert_array_equal(ul[sel], g['u'])
    np.testing.assert_array_equal(vl[sel], g['v'])
    np.testing.assert_array_equal(pl[sel], g['p'])


# ------------------------------------------------------------------ chorin_spectral (a10-a11)
@pytest.mark.parametrize('N', [9, 17, 33, 51])
def test_chorin_spectral_matrices(N):
    g = load_golden('chorin_spectral_%d.npz' % N)
    np.testing.assert_array_equal(OS.gauss_lobatto_points(N), g['x_i'])
    np.testing.assert_array_equal(OS.T_matrix(N), g['Tx'])
    np.testing.assert_allclose(OS.inv_T_matrix(N), g['Tx_inv'], rtol=1e-14, atol=0)
    assert rel_l2(OS.D_matrix(N), g['Dx']) < 1e-14
    assert rel_l2(OS.D_sqr_matrix(N), g['Dx_sqr']) < 1e-13
    assert rel_l2(OS.D_matrix_degrees_minus_2(N), g['DPx']) < 1e-14
    S = OS.Setup(N, N, unpack_bcs(g, 'u_bc'), unpack_bcs(g, 'v_bc'))
    assert rel_l2(S.DxDPx, g['DxDPx']) < 1e-13


@pytest.mark.parametrize('N', [17, 51])
def test_chorin_spectral_single_step(N):
    g = load_golden('chorin_spectral_%d.npz' % N)
    dt, rho = g['params']
    S = OS.Setup(N, N, unpack_bcs(g, 'u_bc'), unpack_bcs(g, 'v_bc'))
    ui, vi = OS.predictor_step(S, g['un'], g['vn'], g['un1'], g['vn1'], dt)
    assert np.isrealobj(ui)
    assert rel_l2(ui, g['pred_ui']) < 1e-8 and rel_l2(vi, g['pred_vi']) < 1e-8
    a, b, c = OS.correction_step(S, g['pred_ui'], g['pred_vi'], g['p'], dt, rho)
    assert rel_l2(a, g['corr_u']) < 1e-8 and rel_l2(b, g['corr_v']) < 1e-8 and rel_l2(c, g['corr_p']) < 1e-8


def test_corrected_options_known_answers():
    """SURVEY section 8 (f) rank 3 options.  (1) Corrected explicit predictor: for u = sin(y), v = 1 (uniform in x) the
    exact advection term v u_y = cos(y) is reproduced to 2nd order, while the reference's form (x-difference twice)
    gives zero.  (2) Red-black SOR converges to the same discrete Poisson solution as the lexicographic sweep."""
    from oracle import chorin_fd as O
    n = 64
    y = np.linspace(0, 2 * np.pi, n)
    dx = dy = y[1] - y[0]
    u = np.tile(np.sin(y)[None, :], (n, 1)); v = np.ones((n, n))
    dt, nu = 1e-3, 0.0
    ui_c, _ = O.explicit_predictor_corrected(u, v, u, v, dt, dx, dy, nu)
    ui_r, _ = O.explicit_predictor(u, v, u, v, dt, dx, dy, nu)
    adv_c = -(ui_c - u)[1:-1, 1:-1] / dt
    assert np.abs(adv_c - np.cos(y)[None, 1:-1]).max() < 2e-3            # O(dy^2)
    assert np.abs((ui_r - u)[1:-1, 1:-1]).max() == 0.0                      # the reference cannot see d/dy
    rng = np.random.default_rng(0)
    nx, ny = 24, 20
    ui, vi = rng.standard_normal((nx, ny)), rng.standard_normal((nx, ny))
    p1 = O.get_pressure(ui, vi, np.zeros((nx, ny)), 1e-2, 0.1, 0.1, 1.0, 1.5, 5000, tol=1e-13)
    p2 = O.get_pressure_redblack(ui, vi, np.zeros((nx, ny)), 1e-2, 0.1, 0.1, 1.0, 1.5, 5000, tol=1e-13)
    assert np.abs(p1 - p2).max() < 1e-9 * max(1.0, np.abs(p1).max())


def test_corrected_adi_transposition_identity():
    """semi_implicit_predictor_corrected: (1) its second solve really runs along y -- for data that are constant along x
    the x-solve input is uniform in x and the result equals an independent 1-D tridiagonal solve along y done with
    scipy; (2) on a square grid with dx == dy it differs from the reference's predictor (which solves along x twice)."""
    from scipy.linalg import solve_banded
    from oracle import chorin_fd as O
    rng = np.random.default_rng(2)
    nx, ny = 12, 20
    dt, dx, dy, nu = 1e-2, 0.1, 0.07, 0.3
    u = rng.standard_normal((nx, ny)); v = rng.standard_normal((nx, ny))
    ui, vi = O.semi_implicit_predictor_corrected(u, v, 0.9 * u, 0.8 * v, dt, dx, dy, nu)
    assert ui.shape == (nx, ny)
    # rebuild ut with the same first step, then solve the second system row by row with an independent solver
    c = (slice(1, -1), slice(1, -1))
    a_diag = 2 / nu * dx**2 + 2 * dt; b_diag = 2 / nu * dy**2 + 2 * dt
    def H(a, b, f): return a[c] * (f[2:, 1:-1] - f[:-2, 1:-1]) / (2 * dx) + b[c] * (f[1:-1, 2:] - f[1:-1, :-2]) / (2 * dy)
    C1 = dt / 2. * (3 * H(u, v, u) - H(0.9 * u, 0.8 * v, 0.9 * u))
    C2 = dt * nu * ((u[2:, 1:-1] - 2 * u[c] + u[:-2, 1:-1]) / dx**2 + (u[1:-1, 2:] - 2 * u[c] + u[1:-1, :-2]) / dy**2)
    n1 = nx - 2
    ab = np.zeros((3, n1)); ab[0, 1:] = -dt; ab[1] = a_diag; ab[2, :-1] = -dt
    ut = solve_banded((1, 1), ab, 2 / nu * dx**2 * (C1 + C2))
    S = 2 / nu * dy**2 * (ut + u[c]) - dt * (u[1:-1, 2:] - 2 * u[c] + u[1:-1, :-2])
    n2 = ny - 2
    ab2 = np.zeros((3, n2)); ab2[0, 1:] = -dt; ab2[1] = b_diag; ab2[2, :-1] = -dt
    ref = solve_banded((1, 1), ab2, S.T).T
    assert np.abs(ui[c] - ref).max() < 1e-12
    assert np.array_equal(ui[0], u[0]) and np.array_equal(ui[:, -1], u[:, -1])
    n = 16
    u = rng.standard_normal((n, n)); v = rng.standard_normal((n, n))
    a, _ = O.semi_implicit_predictor_corrected(u, v, u, v, dt, 0.1, 0.1, nu)
    b, _ = O.semi_implicit_predictor(u, v, u, v, dt, 0.1, 0.1, nu)
    assert np.abs(a - b).max() > 1e-6


@pytest.mark.parametrize('N', [9, 17, 33])
def test_chorin_spectral_corrected_matrices_known_answers(N):
    """The corrected constructors (an option of the build): exact differentiation of every polynomial the N nodes carry,
    D^2 = second derivative, T^-1 T = I -- none of which the reference's matrices satisfy (they stay the default)."""
    x = OS.gauss_lobatto_points(N)
    D, D2 = OS.D_matrix(N, corrected=True), OS.D_sqr_matrix(N, corrected=True)
    for k in range(1, N):
        assert np.abs(D @ x**k - k * x**(k - 1)).max() < 1e-10 * k * k
    for k in range(2, N):
        assert np.abs(D2 @ x**k - k * (k - 1) * x**(k - 2)).max() < 1e-9 * k**4
    assert np.abs(OS.inv_T_matrix(N, corrected=True) @ OS.T_matrix(N) - np.eye(N)).max() < 1e-13
    assert np.abs(OS.D_matrix(N) @ x - 1).max() > 0.1 and np.abs(OS.inv_T_matrix(N) @ OS.T_matrix(N) - np.eye(N)).max() > 0.5
    # one predictor step = the Crank-Nicolson heat step of an eigenfunction (advection is O(eps^2))
    h = 2. / N
    bc = [('dirichlet', s, 0.0, h, h) for s in ('left', 'right', 'top', 'bottom')]
    S = OS.Setup(N, N, bc, bc, corrected=True)
    f = np.sin(np.pi * S.x_i[:, None]) * np.sin(np.pi * S.y_i[None, :])
    eps, dt = 1e-7, 1e-3
    ui, _ = OS.predictor_step(S, eps * f, eps * f, eps * f, eps * f, dt)
    err = np.abs(ui / eps - (2 - 2 * np.pi**2 * dt) / (2 + 2 * np.pi**2 * dt) * f).max()
    assert err < (1e-8 if N >= 17 else 1e-3)                               # spectral accuracy: N = 9 resolves sin(pi x) to ~1e-5
