"""GPU parity of the chorin_spectral mirror (Chebyshev matrices on the host, per-step GEMMs + fused
elementwise kernels on the device) vs golden vectors captured from the reference.  eig ordering /
normalisation is LAPACK-dependent but the products P diag P^-1 are not: 1e-8 rel-L2."""
import numpy as np
import pytest
import torch

from conftest import load_golden, unpack_bcs, rel_l2

pytestmark = pytest.mark.gpu


def objs(bcs):
    from src.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
    return [(D if k == 'dirichlet' else N)(v, s, dx, dy) for (k, s, v, dx, dy) in bcs]


def test_cheb_gemm_all_transposes_and_accumulate(gpu_device):
    from nns import ops
    rng = np.random.default_rng(0)
    for (M, N, K) in ((49, 49, 49), (7, 33, 50), (1, 15, 31), (17, 1, 5)):
        for ta in (False, True):
            for tb in (False, True):
                A = rng.standard_normal((K, M) if ta else (M, K))
                B = rng.standard_normal((N, K) if tb else (K, N))
                C0 = rng.standard_normal((M, N))
                ref = 0.7 * (A.T if ta else A) @ (B.T if tb else B) - 1.3 * C0
                C = torch.as_tensor(C0, device='cuda')
                ops.cheb_gemm(torch.as_tensor(A, device='cuda'), torch.as_tensor(B, device='cuda'), ta, tb, 0.7, -1.3, out=C)
                assert rel_l2(C.cpu().numpy(), ref) < 1e-14


@pytest.mark.parametrize('N', [9, 17, 33, 51])
def test_matrices_match_reference(N, gpu_device):
    from src.chorin_spectral.simulate import NavierStokesSystem
    g = load_golden('chorin_spectral_%d.npz' % N)
    dt, rho = [float(x) for x in g['params']]
    s = NavierStokesSystem(None, None, None, objs(unpack_bcs(g, 'u_bc')), objs(unpack_bcs(g, 'v_bc')), nt=1, nit=1, nx=N, ny=N, dt=dt, rho=rho, nu=0.1)
    for name in ('x_i', 'Dx', 'Dx_sqr', 'DPx', 'DxDPx', 'Tx', 'Tx_inv'):
        assert rel_l2(getattr(s, name), g[name]) < 1e-13, name


@pytest.mark.parametrize('N', [17, 51])
def test_predictor_and_correction_step(N, gpu_device):
    from src.chorin_spectral.simulate import NavierStokesSystem
    g = load_golden('chorin_spectral_%d.npz' % N)
    dt, rho = [float(x) for x in g['params']]
    s = NavierStokesSystem(None, None, None, objs(unpack_bcs(g, 'u_bc')), objs(unpack_bcs(g, 'v_bc')), nt=1, nit=1, nx=N, ny=N, dt=dt, rho=rho, nu=0.1)
    ui, vi = s._predictor_step(g['un'], g['vn'], g['un1'], g['vn1'])
    assert rel_l2(ui, g['pred_ui']) < 1e-8 and rel_l2(vi, g['pred_vi']) < 1e-8
    a, b, c = s._correction_step(g['pred_ui'], g['pred_vi'], g['p'])
    assert rel_l2(c, g['corr_p']) < 1e-8
    # The reference's Uzawa operator is near-singular: Q (= p interior) is ~1e17 here, and the velocity update
    # u -= DxDPx @ Q dt/rho cancels 1e17-size terms down to ~1e3 (this is the divergence SURVEY.md 8c records).
    # A one-ulp change of Q therefore moves u by O(10): u, v can only be held to the forward-error bound of that
    # product, |du| <= gamma_n * (|DxDPx| @ |Q|) * dt/rho, not to 1e-8.
    Q = g['corr_p'][1:-1, 1:-1]
    n = Q.shape[0]
    bu = 8 * n * np.finfo(np.float64).eps * (np.abs(s.DxDPx) @ np.abs(Q)).max() * dt / rho
    bv = 8 * n * np.finfo(np.float64).eps * (np.abs(Q) @ np.abs(s.DyDPy).T).max() * dt / rho
    assert np.abs(a - g['corr_u']).max() <= bu and np.abs(b - g['corr_v']).max() <= bv
    np.testing.assert_array_equal(a[[0, -1]], g['corr_u'][[0, -1]])            # boundary rows are copied from u*
    a2, b2, c2 = s.step(g['un'], g['vn'], g['un1'], g['vn1'], g['p'])
    assert rel_l2(c2, g['corr_p']) < 1e-6


def test_error_behaviour(gpu_device):
    from src.chorin_spectral.simulate import NavierStokesSystem
    from src.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as Nm
    d = 0.1
    good = [D(0, 'left', d, d), D(1, 'right', d, d), D(0, 'top', d, d), D(0, 'bottom', d, d)]
    with pytest.raises(NotImplementedError):                        # Neumann unsupported (:218-221)
        NavierStokesSystem(None, None, None, [Nm(0, 'left', d, d)] + good[1:], good, nx=9, ny=9)
    with pytest.raises(FloatingPointError):                         # complex eigenvalues at N >= 52
        NavierStokesSystem(None, None, None, good, good, nx=64, ny=64)
    z = np.zeros((9, 9))
    ul, vl, pl = NavierStokesSystem(z, z.copy(), z.copy(), good, good, nt=2, nx=9, ny=9).simulate()
    assert ul.shape == (2, 9, 9) and ul.dtype == np.float64


@pytest.mark.parametrize('N', [17, 33])
def test_corrected_matrices_option(N, gpu_device):
    """matrices='corrected' (SURVEY section 8 (f) rank 3): the mirror's matrices equal the oracle's corrected constructors,
    D differentiates every polynomial the nodes carry, and one predictor step on the GPU reproduces the Crank-Nicolson
    heat step of an eigenfunction ((2 - dt lap) u* = (2 + dt lap) u, tiny amplitude so that advection is O(eps^2)) to
    1e-8 -- which the reference's matrices miss by O(10): the default still reproduces the reference (tests above)."""
    from src.chorin_spectral.simulate import NavierStokesSystem
    from src.boundary import DirichletBoundaryCondition as D
    from oracle import chorin_spectral as OS
    h = 2. / N
    bcs = [D(0.0, s, h, h) for s in ('left', 'right', 'top', 'bottom')]
    dt = 1e-3
    s = NavierStokesSystem(None, None, None, bcs, bcs, nt=1, nit=1, nx=N, ny=N, dt=dt, rho=1.0, nu=1.0, matrices='corrected')
    assert rel_l2(s.Dx, OS.D_matrix(N, corrected=True)) < 1e-14 and rel_l2(s.Dx_sqr, OS.D_sqr_matrix(N, corrected=True)) < 1e-14
    assert np.abs(s.Tx_inv @ s.Tx - np.eye(N)).max() < 1e-13
    x = s.x_i
    for k in range(1, N):
        assert np.abs(s.Dx @ x**k - k * x**(k - 1)).max() < 1e-9 * max(1, k)**2
    f = np.sin(np.pi * x[:, None]) * np.sin(np.pi * s.y_i[None, :])
    eps = 1e-7
    fac = (2 - 2 * np.pi**2 * dt) / (2 + 2 * np.pi**2 * dt)
    ui, vi = s._predictor_step(eps * f, eps * f, eps * f, eps * f)
    assert np.abs(ui / eps - fac * f).max() < 1e-8 and np.abs(vi / eps - fac * f).max() < 1e-8
    ref = NavierStokesSystem(None, None, None, bcs, bcs, nt=1, nit=1, nx=N, ny=N, dt=dt, rho=1.0, nu=1.0)
    ur, _ = ref._predictor_step(eps * f, eps * f, eps * f, eps * f)
    assert np.abs(ur / eps - fac * f).max() > 1.0                          # the reference's matrices do not solve this problem
    # the corrected path on the GPU == the oracle with the corrected constructors
    S = OS.Setup(N, N, [('dirichlet', b.boundary, 0.0, h, h) for b in bcs], [('dirichlet', b.boundary, 0.0, h, h) for b in bcs], corrected=True)
    rng = np.random.default_rng(3)
    fields = [0.1 * rng.standard_normal((N, N)) for _ in range(4)]
    got = s._predictor_step(*fields)
    want = OS.predictor_step(S, *fields, dt)
    assert rel_l2(got[0], want[0]) < 1e-8 and rel_l2(got[1], want[1]) < 1e-8


def _bc_objs(kinds, values, h):
    from src.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as Nm
    return [(Nm if k == 'neumann' else D)(v, s, h, h) for k, v, s in zip(kinds, values, ('left', 'right', 'top', 'bottom'))]


@pytest.mark.parametrize('N', [17, 33])
def test_corrected_path_neumann_and_inhomogeneous_data(N, gpu_device):
    """matrices='corrected' accepts Neumann data (the reference raises NotImplementedError, :218-221) and
    folds the boundary values into every derivative.  Analytic answers of one predictor step, (2 - dt lap) u* = (2 + dt lap) u
    at tiny amplitude: Neumann and mixed heat eigenfunctions decay by the Crank-Nicolson factor, harmonic profiles carried by
    inhomogeneous Dirichlet / Neumann data stay put; and the GPU path equals the oracle's corrected path."""
    from src.chorin_spectral.simulate import NavierStokesSystem
    from oracle import chorin_spectral as OS
    h, dt, eps = 2. / N, 1e-3, 1e-7
    fac = (2 - 2 * np.pi**2 * dt) / (2 + 2 * np.pi**2 * dt)
    mk = lambda kinds, values: NavierStokesSystem(None, None, None, _bc_objs(kinds, values, h), _bc_objs(kinds, values, h), nt=1, nit=1, nx=N, ny=N,
                                                  dt=dt, rho=1.0, nu=1.0, matrices='corrected')
    s = mk(['neumann'] * 4, [0.0] * 4)
    x, y = s.x_i[:, None], s.y_i[None, :]
    inner = lambda a: a[1:-1, 1:-1]
    f = np.cos(np.pi * x) * np.cos(np.pi * y)                                    # df/dn = 0 on every side
    ui, _ = s._predictor_step(eps * f, eps * f, eps * f, eps * f)
    assert np.abs(inner(ui / eps - fac * f)).max() < 1e-8
    assert np.abs((ui / eps - fac * f)[0, 1:-1]).max() < 1e-8 and np.abs((ui / eps - fac * f)[1:-1, -1]).max() < 1e-8      # the walls move too
    s = mk(['dirichlet', 'dirichlet', 'neumann', 'neumann'], [0.0] * 4)
    f = np.sin(np.pi * x) * np.cos(np.pi * y)
    ui, _ = s._predictor_step(eps * f, eps * f, eps * f, eps * f)
    assert np.abs(inner(ui / eps - fac * f)).max() < 1e-8
    s = mk(['dirichlet', 'dirichlet', 'neumann', 'neumann'], [0.0, eps, 0.0, 0.0])        # u = 0 at x = -1 ('left'), eps at x = +1
    f = eps * (1 + x) / 2 * np.ones_like(y)
    ui, _ = s._predictor_step(f, f, f, f)
    assert np.abs(ui - f)[:, 1:-1].max() < 1e-8 * eps * 10
    s = mk(['neumann'] * 4, [eps, eps, 0.0, 0.0])                                           # du/dx = eps on both x sides
    f = eps * x * np.ones_like(y)
    ui, _ = s._predictor_step(f, f, f, f)
    assert np.abs(ui - f)[:, 1:-1].max() < 1e-8 * eps * 10
    with pytest.raises(NotImplementedError):                                                 # the default still mirrors the reference
        NavierStokesSystem(None, None, None, _bc_objs(['neumann'] * 4, [0.0] * 4, h), _bc_objs(['dirichlet'] * 4, [0.0] * 4, h), nx=N, ny=N)
    # GPU == oracle on random fields with mixed, inhomogeneous data
    kinds, vals = ['neumann', 'dirichlet', 'dirichlet', 'neumann'], [0.3, -0.2, 0.1, 0.4]
    s = mk(kinds, vals)
    tup = [(k, sd, v, h, h) for k, v, sd in zip(kinds, vals, ('left', 'right', 'top', 'bottom'))]
    S = OS.Setup(N, N, tup, tup, corrected=True)
    rng = np.random.default_rng(N)
    fields = [0.1 * rng.standard_normal((N, N)) for _ in range(4)]
    got, want = s._predictor_step(*fields), OS.predictor_step(S, *fields, dt)
    assert rel_l2(got[0], want[0]) < 1e-8 and rel_l2(got[1], want[1]) < 1e-8


@pytest.mark.parametrize('N', [17, 33])
def test_corrected_projection_is_well_conditioned(N, gpu_device):
    """The corrected correction step (exact interior pressure derivative, constant pressure mode projected out, gradient update):
    a TIGHT check of a11's second half, which the reference's near-singular operator only allows to a forward-error bound --
    u* = (dt / rho) grad(phi) with no normal flux through the walls is projected to zero velocity and p = phi (up to a constant);
    random data lose all of their interior divergence but the incompatible constant; GPU == oracle to 1e-9."""
    from src.chorin_spectral.simulate import NavierStokesSystem
    from oracle import chorin_spectral as OS
    h, dt, rho = 2. / N, 1e-3, 1.3
    bcs = _bc_objs(['dirichlet'] * 4, [0.0] * 4, h)
    s = NavierStokesSystem(None, None, None, bcs, bcs, nt=1, nit=1, nx=N, ny=N, dt=dt, rho=rho, nu=1.0, matrices='corrected')
    lam = np.sort(np.abs(s.DxDPx_lambda))
    assert lam[0] < 1e-10 and abs(lam[1] - np.pi**2 / 4) < 1e-6                  # the Neumann Laplacian's spectrum: 0, (pi/2)^2, ...
    x, y = s.x_i[:, None], s.y_i[None, :]
    phi = np.cos(np.pi * x) * np.cos(np.pi * y)
    ui = dt / rho * (-np.pi * np.sin(np.pi * x) * np.cos(np.pi * y))
    vi = dt / rho * (-np.pi * np.cos(np.pi * x) * np.sin(np.pi * y))
    u1, v1, p1 = s._correction_step(ui, vi, np.zeros((N, N)))
    Q, ph = p1[1:-1, 1:-1], phi[1:-1, 1:-1]
    assert np.abs(u1[1:-1, 1:-1]).max() < 1e-8 * np.abs(ui).max() and np.abs(v1[1:-1, 1:-1]).max() < 1e-8 * np.abs(vi).max()
    assert np.abs((Q - Q.mean()) - (ph - ph.mean())).max() < 1e-8
    tup = [('dirichlet', sd, 0.0, h, h) for sd in ('left', 'right', 'top', 'bottom')]
    S = OS.Setup(N, N, tup, tup, corrected=True)
    rng = np.random.default_rng(N + 1)
    ui, vi, p = (rng.standard_normal((N, N)) for _ in range(3))
    got, want = s._correction_step(ui, vi, p), OS.correction_step_corrected(S, ui, vi, p, dt, rho)
    for a, b in zip(got, want):
        assert rel_l2(a, b) < 1e-9
    div = s.Dx[1:-1, :] @ got[0][:, 1:-1] + got[1][1:-1, :] @ s.Dy[1:-1, :].T
    div0 = s.Dx[1:-1, :] @ ui[:, 1:-1] + vi[1:-1, :] @ s.Dy[1:-1, :].T
    assert np.abs(div - div.mean()).max() < 1e-9 * np.abs(div0).max()


@pytest.mark.parametrize('matrices', ['reference', 'corrected'])
def test_simulate_replayed_from_a_hip_graph_is_bitwise_the_eager_loop(matrices, gpu_device):
    """Round 4: `simulate` captures ONE step (the ~90 launches of predictor + correction on N x N matrices) as a HIP graph and replays it; the
    trajectories must be bitwise those of the eager loop -- same kernels in the same order on the same buffers."""
    from src.chorin_spectral.simulate import NavierStokesSystem
    from src.boundary import DirichletBoundaryCondition as D
    N = 17
    h = 2. / N
    lid = [D(0.0, 'left', h, h), D(0.0, 'right', h, h), D(1.0, 'top', h, h), D(0.0, 'bottom', h, h)]
    wall = [D(0.0, s, h, h) for s in ('left', 'right', 'top', 'bottom')]
    rng = np.random.default_rng(4)
    ic = [1e-3 * rng.standard_normal((N, N)) for _ in range(3)]
    runs = []
    for g in (True, False):
        s = NavierStokesSystem(ic[0].copy(), ic[1].copy(), ic[2].copy(), lid, wall, nt=8, nit=1, nx=N, ny=N, dt=1e-4, rho=1.0, nu=1.0, matrices=matrices)
        runs.append(s.simulate(use_graph=g))
        assert s.last_simulate_used_graph == g                       # (a capture that fails falls back to the eager loop: that must not be what passes here)
    for a, b in zip(*runs):
        assert a.shape == (8, N, N) and np.isfinite(a).all() and np.array_equal(a, b)
