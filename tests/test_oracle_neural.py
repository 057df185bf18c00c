"""The neural oracle (oracle/neural.py) vs golden vectors captured from the reference's PDEFunc /
ANODE integrators / BasisFunc (float32 in the reference; the oracle is run in float32 here for a
tight comparison and in float64 to bound the fp32 rounding of the reference itself)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import neural as ON

G = load_golden('neural_spectral.npz')


def T(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype)


def mlp_from(prefix, dtype=torch.float32, req=False):
    ps = [T(G[prefix + 'net.%d.%s' % (i, w)], dtype) for i in (0, 2, 4) for w in ('weight', 'bias')]
    if req:
        for p in ps:
            p.requires_grad_(True)
    return tuple(ps)


@pytest.mark.parametrize('method', ['Euler', 'RK2', 'RK4'])
def test_integrators(method):
    out = ON.integrate(mlp_from('ode_'), T(G['ode_z0']), 7, method)
    assert out.shape == (7, 3, 12)
    assert rel_l2(out.numpy(), G['ode_' + method]) < 2e-6
    out64 = ON.integrate(mlp_from('ode_', torch.float64), T(G['ode_z0'], torch.float64), 7, method)
    assert rel_l2(out64.numpy(), G['ode_' + method]) < 2e-6


@pytest.mark.parametrize('mb', [1, 3])
def test_spectral_ode_forward_loss_grads(mb):
    K = 4
    init = T(G['s1_param_init_coeffs']).requires_grad_(True)
    mlp = mlp_from('s1_param_basis_coeffs.', req=True)
    basis = torch.stack([T(G['s1_param_basis_fns.%d' % k]) for k in range(K)]).requires_grad_(True)
    obs = T(G['s1_mb%d_obs' % mb])
    pred, _ = ON.pde_forward(init, mlp, basis, mb, obs.shape[0])
    assert rel_l2(pred.detach().numpy(), G['s1_mb%d_pred' % mb]) < 2e-6
    loss = ON.loss_fn(pred, obs)
    assert abs(loss.item() - float(G['s1_mb%d_loss' % mb])) < 1e-5 * float(G['s1_mb%d_loss' % mb])
    loss.backward()
    pre = 's1_mb%d_grad_' % mb
    assert rel_l2(init.grad.numpy(), G[pre + 'init_coeffs']) < 1e-4
    for i, idx in enumerate((0, 2, 4)):
        assert rel_l2(mlp[2 * i].grad.numpy(), G[pre + 'basis_coeffs.net.%d.weight' % idx]) < 1e-4
        assert rel_l2(mlp[2 * i + 1].grad.numpy(), G[pre + 'basis_coeffs.net.%d.bias' % idx]) < 1e-4
    for k in range(K):
        assert rel_l2(basis.grad[k].numpy(), G[pre + 'basis_fns.%d' % k]) < 1e-5


@pytest.mark.parametrize('mb', [1, 3])
def test_spectral_ode2_forward_loss_grads(mb):
    K = 4
    inits = [T(G['s2_param_%s_init_coeffs' % c]).requires_grad_(True) for c in 'uvp']
    mlps = [mlp_from('s2_param_%s_basis_coeffs.' % c, req=True) for c in 'uvp']
    bases = [torch.stack([T(G['s2_param_%s_basis_fns.%d' % (c, k)]) for k in range(K)]).requires_grad_(True)
             for c in 'uvp']
    obs = T(G['s2_mb%d_obs' % mb])
    pred = ON.pde2_forward(inits, mlps, bases, mb, obs.shape[0])
    assert rel_l2(pred.detach().numpy(), G['s2_mb%d_pred' % mb]) < 2e-6
    loss = ON.loss_fn(pred, obs)
    loss.backward()
    pre = 's2_mb%d_grad_' % mb
    for ci, c in enumerate('uvp'):
        assert rel_l2(inits[ci].grad.numpy(), G[pre + '%s_init_coeffs' % c]) < 1e-4
        for k in range(K):
            assert rel_l2(bases[ci].grad[k].numpy(), G[pre + '%s_basis_fns.%d' % (c, k)]) < 1e-5
        assert rel_l2(mlps[ci][2].grad.numpy(), G[pre + '%s_basis_coeffs.net.2.weight' % c]) < 1e-4


def test_diversity_penalty():
    basis = torch.stack([T(G['s1_param_basis_fns.%d' % k]) for k in range(4)])
    assert abs(ON.diversity_penalty(basis).item() - float(G['s1_diversity_penalty'])) < 1e-6 * float(
        G['s1_diversity_penalty'])


def test_pixel_mlp_basisfunc():
    idx = (0, 2, 4, 6, 8)
    Ws = [T(G['bf_param_net.%d.weight' % i])[:, :, 0, 0].clone().requires_grad_(True) for i in idx]
    bs = [T(G['bf_param_net.%d.bias' % i]).requires_grad_(True) for i in idx]
    x = T(G['bf_in']).requires_grad_(True)
    y = ON.pixel_mlp(Ws, bs, x)
    assert rel_l2(y.detach().numpy(), G['bf_out']) < 2e-6
    (y * T(G['bf_w'])).sum().backward()
    assert rel_l2(x.grad.numpy(), G['bf_grad_in']) < 1e-5
    for W, b, i in zip(Ws, bs, idx):
        assert rel_l2(W.grad.numpy(), G['bf_grad_net.%d.weight' % i][:, :, 0, 0]) < 1e-5
        assert rel_l2(b.grad.numpy(), G['bf_grad_net.%d.bias' % i]) < 1e-5


def test_pixel_mlp_backward_matches_autograd():
    """The hand-written reverse mode used as the checker of the fused HIP backward equals torch.autograd (float64)."""
    torch.manual_seed(3)
    dims = [3, 16, 24, 5]
    Ws = [torch.randn(dims[i + 1], dims[i], dtype=torch.float64, requires_grad=True) for i in range(3)]
    bs = [torch.randn(dims[i + 1], dtype=torch.float64, requires_grad=True) for i in range(3)]
    x = torch.randn(2, 3, 5, 7, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(2, 5, 5, 7, dtype=torch.float64)
    y = ON.pixel_mlp(Ws, bs, x)
    y.backward(gy)
    gx, gWs, gbs = ON.pixel_mlp_backward([w.detach() for w in Ws], [b.detach() for b in bs], x.detach(), gy)
    assert torch.allclose(gx, x.grad, rtol=1e-12, atol=1e-12)
    for l in range(3):
        assert torch.allclose(gWs[l], Ws[l].grad, rtol=1e-12, atol=1e-12)
        assert torch.allclose(gbs[l], bs[l].grad, rtol=1e-12, atol=1e-12)
