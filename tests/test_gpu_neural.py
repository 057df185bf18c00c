"""GPU parity of the neural_spectral predictor (fused RK4-MLP kernel, basis expansion, fused loss and
their hand-written backward kernels) against golden vectors captured from the reference's PDEFunc /
ANODE code and against the float64 torch-CPU oracle.  The reference computes in float32, so goldens
themselves carry ~1e-6 rounding: forward <= 1e-5 rel-L2, gradients <= 2e-4 (same bar the oracle is held to
in tests/test_oracle_neural.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
G = load_golden('neural_spectral.npz')


def T(a, dtype=torch.float32, device='cuda'):
    return torch.tensor(np.asarray(a), dtype=dtype, device=device)


def load_model(cls, prefix, K, nx, ny):
    m = cls(K, nx, ny)
    sd = {k[len(prefix):]: T(G[k], device='cpu') for k in G.files if k.startswith(prefix)}
    m.load_state_dict(sd, strict=True)                       # parameter-name compatibility with the reference
    return m.cuda()


def oracle_mlp(mod, dtype=torch.float64):
    return tuple(p.detach().cpu().to(dtype).clone().requires_grad_(True)
                 for p in (mod.net[0].weight, mod.net[0].bias, mod.net[2].weight, mod.net[2].bias, mod.net[4].weight, mod.net[4].bias))


@pytest.mark.parametrize('method', ['Euler', 'RK2', 'RK4'])
def test_integrators_vs_golden_and_grads_vs_oracle(method, gpu_device):
    from src.neural_spectral.spectral_ode import ODEFunc
    from src.neural_spectral.anode import odesolver, odesolver_adjoint
    from oracle import neural as ON
    f = ODEFunc(12)
    f.load_state_dict({k[len('ode_'):]: T(G[k], device='cpu') for k in G.files if k.startswith('ode_net')})
    f = f.cuda()
    z0 = T(G['ode_z0']).requires_grad_(True)
    out = odesolver_adjoint(f, z0, {'Nt': 7, 'method': method})
    assert out.shape == (7, 3, 12)
    assert rel_l2(out.detach().cpu().numpy(), G['ode_' + method]) < 1e-5
    assert torch.equal(out, odesolver(f, z0, {'Nt': 7, 'method': method}))
    w = torch.randn(7, 3, 12, generator=torch.Generator().manual_seed(1))
    (out * w.cuda()).sum().backward()
    mlp = oracle_mlp(f)
    z64 = T(G['ode_z0'], torch.float64, 'cpu').requires_grad_(True)
    (ON.integrate(mlp, z64, 7, method) * w.double()).sum().backward()
    assert rel_l2(z0.grad.cpu().numpy(), z64.grad.numpy()) < 2e-5
    for got, ref in zip((f.net[0].weight, f.net[0].bias, f.net[2].weight, f.net[2].bias, f.net[4].weight, f.net[4].bias), mlp):
        assert rel_l2(got.grad.cpu().numpy(), ref.grad.numpy()) < 2e-5


def test_ode_kernel_many_rows_and_long_horizon(gpu_device):
    """mb > 16 (several batch tiles, atomics for the weight gradients), ragged mb, K = 30, Nt = 100 (cfg 2)."""
    from nns.neural_spectral.spectral_ode import ODEFunc
    from nns.neural_spectral.anode import odesolver
    from oracle import neural as ON
    torch.manual_seed(3)
    f = ODEFunc(30).cuda()
    z0 = torch.randn(37, 30, device='cuda', requires_grad=True)
    out = odesolver(f, z0, {'Nt': 100, 'method': 'RK4'})
    w = torch.randn(100, 37, 30, generator=torch.Generator().manual_seed(2))
    (out * w.cuda()).sum().backward()
    mlp = oracle_mlp(f)
    z64 = z0.detach().cpu().double().requires_grad_(True)
    ref = ON.integrate(mlp, z64, 100, 'RK4')
    (ref * w.double()).sum().backward()
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert rel_l2(z0.grad.cpu().numpy(), z64.grad.numpy()) < 1e-4
    for got, r in zip((f.net[0].weight, f.net[0].bias, f.net[2].weight, f.net[2].bias, f.net[4].weight, f.net[4].bias), mlp):
        assert rel_l2(got.grad.cpu().numpy(), r.grad.numpy()) < 1e-4
    with pytest.raises(TypeError):
        odesolver(f, z0.detach(), None)                     # reference: options=None -> TypeError at options['method']


@pytest.mark.parametrize('mb', [1, 3])
@pytest.mark.parametrize('fused', [False, True])
def test_spectral_ode_pdefunc_vs_golden(mb, fused, gpu_device):
    from src.neural_spectral.spectral_ode import PDEFunc
    m = load_model(PDEFunc, 's1_param_', 4, 16, 16)
    obs = T(G['s1_mb%d_obs' % mb])
    t = torch.arange(obs.shape[0], device='cuda') + 1
    if fused:
        loss = m.loss(obs[0], t, obs)
    else:
        pred = m(obs[0], t)
        assert pred.shape == (obs.shape[0], mb, 3, 16, 16)
        assert rel_l2(pred.detach().cpu().numpy(), G['s1_mb%d_pred' % mb]) < 1e-5
        loss = torch.norm(pred - obs, p=2)
    ref_loss = float(G['s1_mb%d_loss' % mb])
    assert abs(loss.item() - ref_loss) < 1e-5 * ref_loss
    loss.backward()
    pre = 's1_mb%d_grad_' % mb
    for name, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), G[pre + name]) < 2e-4, name
    if not fused and mb == 1:
        assert abs(m.diversity_penalty().item() - float(G['s1_diversity_penalty'])) < 1e-5 * float(G['s1_diversity_penalty'])


@pytest.mark.parametrize('mb', [1, 3])
def test_spectral_ode2_pdefunc_vs_golden(mb, gpu_device):
    from src.neural_spectral.spectral_ode2 import PDEFunc
    m = load_model(PDEFunc, 's2_param_', 4, 16, 16)
    obs = T(G['s2_mb%d_obs' % mb])
    t = torch.arange(obs.shape[0], device='cuda') + 1
    pred = m(obs[0], t)
    assert rel_l2(pred.detach().cpu().numpy(), G['s2_mb%d_pred' % mb]) < 1e-5
    loss = torch.norm(pred - obs, p=2)
    assert abs(loss.item() - float(G['s2_mb%d_loss' % mb])) < 1e-5 * float(G['s2_mb%d_loss' % mb])
    loss.backward()
    pre = 's2_mb%d_grad_' % mb
    for name, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), G[pre + name]) < 2e-4, name
    m.zero_grad()
    m.loss(obs[0], t, obs).backward()
    for name, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), G[pre + name]) < 2e-4, name


def test_cfg2_size_fused_training_step_vs_oracle(gpu_device):
    """BASELINE config 2: 128x128 periodic box, K = 10, nt = 100, mb = 1, float32: one fused forward + backward
    against the float64 oracle, and an Adam step runs on the device parameters."""
    from nns.neural_spectral.spectral_ode import PDEFunc
    from oracle import neural as ON
    torch.manual_seed(0)
    K, n, nt, mb = 10, 128, 100, 1
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, mb, 3, n, n, device='cuda')
    t = torch.arange(nt, device='cuda') + 1
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.zero_grad()
    loss = m.loss(obs[0], t, obs)
    loss.backward()
    init = m.init_coeffs.detach().cpu().double().requires_grad_(True)
    mlp = oracle_mlp(m.basis_coeffs)
    basis = torch.stack([f.detach().cpu().double() for f in m.basis_fns]).requires_grad_(True)
    pred, _ = ON.pde_forward(init, mlp, basis, mb, nt)
    ref = ON.loss_fn(pred, obs.cpu().double())
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * ref.item()
    assert rel_l2(m.init_coeffs.grad.cpu().numpy(), init.grad.numpy()) < 1e-4
    assert rel_l2(torch.stack([f.grad for f in m.basis_fns]).cpu().numpy(), basis.grad.numpy()) < 1e-5
    assert rel_l2(m.basis_coeffs.net[2].weight.grad.cpu().numpy(), mlp[2].grad.numpy()) < 1e-4
    before = m.init_coeffs.detach().clone()
    opt.step()
    assert not torch.equal(before, m.init_coeffs.detach())


def test_pixel_mlp_basisfunc_vs_golden_and_deep_bf16(gpu_device):
    """BasisFunc (reference widths 3-16-32-32-16-3) through the fused MFMA kernel vs the reference's output; a
    depth-8 width-64 stack (config 3) in float32 vs the float64 oracle and in bfloat16 at bf16 tolerance."""
    from src.neural_spectral.spectral_ode import BasisFunc
    from nns.neural_spectral.spectral_ode import PixelMLP
    from nns import ops
    from oracle import neural as ON
    bf = BasisFunc(8, 8)
    bf.load_state_dict({k[len('bf_param_'):]: T(G[k], device='cpu') for k in G.files if k.startswith('bf_param_')})
    bf = bf.cuda()
    x = T(G['bf_in'])
    y = bf.fused_forward(x)
    assert y.shape == (2, 3, 8, 8)
    assert rel_l2(y.cpu().numpy(), G['bf_out']) < 1e-5
    assert rel_l2(bf(x).detach().cpu().numpy(), G['bf_out']) < 1e-5          # the unfused torch path, for reference
    assert rel_l2(bf.fused_forward(x, bf16=True).cpu().numpy(), G['bf_out']) < 3e-2
    torch.manual_seed(5)
    for depth, width, shape in ((8, 64, (3, 3, 37, 41)), (4, 32, (1, 3, 128, 128)), (2, 48, (2, 3, 9, 5)), (1, 3, (1, 3, 4, 4))):
        m = PixelMLP(depth, width).cuda()
        for b in m.biases:
            torch.nn.init.normal_(b, std=0.3)
        xx = torch.randn(*shape, device='cuda')
        ref = ON.pixel_mlp([w.detach().cpu().double() for w in m.weights], [b.detach().cpu().double() for b in m.biases], xx.cpu().double()).numpy()
        assert rel_l2(m(xx).cpu().numpy(), ref) < 1e-5, (depth, width)
        assert rel_l2(m(xx, bf16=True).cpu().numpy(), ref) < 5e-2, (depth, width)
    with pytest.raises(Exception):
        PixelMLP(9, 16).cuda()(torch.randn(1, 3, 4, 4, device='cuda'))           # > 8 layers: unsupported, loud
