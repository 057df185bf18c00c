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
