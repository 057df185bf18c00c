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


def test_pixel_mlp_config3_full_shape_properties(gpu_device):
    """BASELINE config 3's field-prediction MLP (depth 8, width 64, bf16 operands) at the shape the bench times it on -- 16 x 512^2 pixels --
    through size-independent properties: run-to-run determinism (bitwise), batch independence (item b of the batch == the item alone,
    bitwise), pixel independence (a permuted copy of 4096 pixels gives the permuted outputs, bitwise), and a 4096-pixel sample against the
    float64 oracle at bf16 tolerance (float32 operands: 1e-5)."""
    from nns.neural_spectral.spectral_ode import PixelMLP
    from oracle import neural as ON
    torch.manual_seed(11)
    m = PixelMLP(8, 64).cuda()
    for b in m.biases:
        torch.nn.init.normal_(b, std=0.3)
    x = torch.randn(16, 3, 512, 512, device='cuda')
    y = m(x, bf16=True)
    assert y.shape == x.shape and bool(torch.isfinite(y).all())
    assert torch.equal(y, m(x, bf16=True))                                            # deterministic
    for b in (0, 7, 15):
        assert torch.equal(y[b:b + 1], m(x[b:b + 1].contiguous(), bf16=True))          # batch independent
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    idx = torch.randint(0, 512 * 512, (4096,), device='cuda', generator=g)
    xs = x[5].reshape(3, -1)[:, idx].reshape(1, 3, 64, 64).contiguous()              # 4096 pixels of item 5, as a 64 x 64 field
    ys = m(xs, bf16=True)
    assert torch.equal(ys.reshape(3, -1), y[5].reshape(3, -1)[:, idx])               # per-pixel operator: position does not matter
    ref = ON.pixel_mlp([w.detach().cpu().double() for w in m.weights], [b.detach().cpu().double() for b in m.biases], xs.cpu().double()).numpy()
    assert rel_l2(ys.cpu().numpy(), ref) < 5e-2
    assert rel_l2(m(xs, bf16=False).cpu().numpy(), ref) < 1e-5


def test_pixel_mlp_backward_exact_integers(gpu_device):
    """Indexing check of the fused backward with data on which bf16 arithmetic is EXACT: sparse weights in {-1, 0, 1},
    integer inputs and upstream gradients, so every operand and every partial sum is a small integer.  Any wrong lane /
    register / k-order in the accumulator chaining or the transposing LDS reads shows up as a non-zero difference."""
    from nns import ops
    from oracle import neural as ON
    g = torch.Generator().manual_seed(11)
    for dims, shape in (([3, 64, 64, 3], (2, 9, 15)), ([3, 32, 48, 16, 3], (1, 16, 16)), ([5, 64, 64, 64, 64, 2], (3, 7, 5)), ([3, 3], (1, 4, 4))):
        L = len(dims) - 1
        Ws = [((torch.rand(dims[i + 1], dims[i], generator=g) < 0.12).float() * (torch.randint(0, 2, (dims[i + 1], dims[i]), generator=g) * 2 - 1).float()) for i in range(L)]
        bs = [torch.randint(-1, 2, (dims[i + 1],), generator=g).float() for i in range(L)]
        x = torch.randint(-2, 3, (shape[0], dims[0]) + shape[1:], generator=g).float()
        gy = torch.randint(-1, 2, (shape[0], dims[-1]) + shape[1:], generator=g).float()
        ref_gx, ref_gW, ref_gb = ON.pixel_mlp_backward([w.double() for w in Ws], [b.double() for b in bs], x.double(), gy.double())
        # keep the case honest: every intermediate must be exactly representable in bf16 (|v| <= 256)
        h = x.double()
        for l in range(L):
            h = torch.einsum('oc,bcxy->boxy', Ws[l].double(), h) + bs[l].double()[None, :, None, None]
            assert h.abs().max() <= 256, (dims, l)
            h = torch.relu(h)
        gx, gW, gb = ops.pixel_mlp_bwd(x.cuda(), gy.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
        assert torch.equal(gx.cpu().double(), ref_gx), dims
        for l in range(L):
            assert torch.equal(gW[l].cpu().double(), ref_gW[l]), (dims, l)
            assert torch.equal(gb[l].cpu().double(), ref_gb[l]), (dims, l)


def test_pixel_mlp_backward_random_and_autograd(gpu_device):
    """Random float data (depth 8 width 64 = BASELINE config 3, ragged pixel counts): 1e-2 against the float64 oracle
    with the kernel's bf16 operand rounding emulated -- what is left (5e-3 at depth 8) are operands that fall on the
    other side of a bf16 rounding boundary, or a ReLU mask that flips, because the kernel accumulates in float32 and the
    oracle exactly; a loose sanity bound against the unrounded oracle; then the autograd node used for training.
    (Indexing is pinned bit-exactly by test_pixel_mlp_backward_exact_integers.)"""
    from nns import ops
    from nns.neural_spectral.spectral_ode import PixelMLP
    from oracle import neural as ON
    torch.manual_seed(7)
    for depth, width, shape in ((8, 64, (3, 3, 37, 41)), (4, 32, (2, 3, 64, 64)), (2, 48, (2, 3, 9, 5))):
        m = PixelMLP(depth, width).cuda()
        for b in m.biases:
            torch.nn.init.normal_(b, std=0.3)
        x = torch.randn(*shape, device='cuda')
        gy = torch.randn(shape[0], 3, *shape[2:], device='cuda')
        args = ([w.detach().cpu().double() for w in m.weights], [b.detach().cpu().double() for b in m.biases], x.cpu().double(), gy.cpu().double())
        ref_gx, ref_gW, ref_gb = ON.pixel_mlp_backward(*args, bf16=True)
        ex_gx, ex_gW, ex_gb = ON.pixel_mlp_backward(*args)
        gx, gW, gb = ops.pixel_mlp_bwd(x, gy, [w.detach() for w in m.weights], [b.detach() for b in m.biases])
        assert rel_l2(gx.cpu().numpy(), ref_gx.numpy()) < 1e-2, (depth, width)
        assert rel_l2(gx.cpu().numpy(), ex_gx.numpy()) < 2e-1, (depth, width)
        for l in range(depth):
            assert rel_l2(gW[l].cpu().numpy(), ref_gW[l].numpy()) < 1e-2, (depth, width, l)
            assert rel_l2(gb[l].cpu().numpy(), ref_gb[l].numpy()) < 1e-2, (depth, width, l)
            assert rel_l2(gW[l].cpu().numpy(), ex_gW[l].numpy()) < 2e-1, (depth, width, l)
    # autograd node: one optimiser step changes the parameters and lowers a simple loss
    m = PixelMLP(4, 32).cuda()
    x = torch.randn(2, 3, 32, 32, device='cuda')
    target = torch.tanh(x.flip(1))
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = ((m.train_forward(x) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0], losses


def test_pixel_mlp_backward_float32(gpu_device):
    """The float32-operand backward (widths <= 32, BASELINE config 2's depth-4 width-32 stack): 2e-5 against the
    UNROUNDED float64 oracle, ragged pixel counts and non-(u, v, p) channel counts, autograd; wider stacks fail loudly."""
    from nns import ops
    from nns.neural_spectral.spectral_ode import PixelMLP
    from oracle import neural as ON
    torch.manual_seed(9)
    for dims, shape in (([3, 32, 32, 32, 3], (2, 64, 64)), ([3, 16, 32, 32, 16, 3], (2, 8, 8)), ([5, 24, 7], (3, 9, 5)), ([3, 3], (1, 4, 4))):
        L = len(dims) - 1
        Ws = [torch.randn(dims[i + 1], dims[i], device='cuda') / dims[i] ** 0.5 for i in range(L)]
        bs = [0.3 * torch.randn(dims[i + 1], device='cuda') for i in range(L)]
        x = torch.randn(shape[0], dims[0], *shape[1:], device='cuda')
        gy = torch.randn(shape[0], dims[-1], *shape[1:], device='cuda')
        ref_gx, ref_gW, ref_gb = ON.pixel_mlp_backward([w.cpu().double() for w in Ws], [b.cpu().double() for b in bs], x.cpu().double(), gy.cpu().double())
        gx, gW, gb = ops.pixel_mlp_bwd(x, gy, Ws, bs, bf16=False)
        assert rel_l2(gx.cpu().numpy(), ref_gx.numpy()) < 2e-5, dims
        for l in range(L):
            assert rel_l2(gW[l].cpu().numpy(), ref_gW[l].numpy()) < 2e-5, (dims, l)
            assert rel_l2(gb[l].cpu().numpy(), ref_gb[l].numpy()) < 2e-5, (dims, l)
    m = PixelMLP(4, 32).cuda()
    x = torch.randn(2, 3, 16, 16, device='cuda', requires_grad=True)
    y = m.train_forward(x, bf16=False)
    y.square().sum().backward()
    ref = ON.pixel_mlp([w.detach().cpu().double().requires_grad_(True) for w in m.weights], [b.detach().cpu().double() for b in m.biases], x.detach().cpu().double())
    assert rel_l2(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5 and x.grad is not None and m.weights[0].grad is not None
    wide = PixelMLP(3, 64).cuda()
    with pytest.raises(Exception):
        ops.pixel_mlp_bwd(x.detach(), torch.randn_like(x), [w.detach() for w in wide.weights], [b.detach() for b in wide.biases], bf16=False)   # > 32: loud


def test_gru_baselines_vs_golden(gpu_device):
    """the GRU variants against outputs captured from the reference
    (tests/golden/neural_rnn.npz).  spectral_rnn.PDEFunc: GRU recurrence on MIOpen, expansion / loss / gradients on the
    fused HIP kernels; rnn.RNN: forward, hidden state and the autoregressive extrapolation."""
    from src.neural_spectral.spectral_rnn import PDEFunc
    from src.neural_spectral.rnn import RNN
    GR = load_golden('neural_rnn.npz')
    K, nx, ny, nt = 4, 8, 8, 6
    m = PDEFunc(K, nx, ny)
    m.load_state_dict({k[len('sr_param_'):]: T(GR[k], device='cpu') for k in GR.files if k.startswith('sr_param_')}, strict=True)
    m = m.cuda()
    for mb in (1, 2):
        pre = 'sr_mb%d_' % mb
        obs = T(GR[pre + 'obs'])
        t = torch.arange(nt, device='cuda') + 1
        pred = m(obs[0], t)
        assert pred.shape == (nt, mb, 3, nx, ny)
        assert rel_l2(pred.detach().cpu().numpy(), GR[pre + 'pred']) < 1e-5
        m.zero_grad()
        loss = m.loss(obs[0], t, obs)                                  # fused loss kernel
        assert abs(loss.item() - float(GR[pre + 'loss'])) < 1e-5 * float(GR[pre + 'loss'])
        loss.backward()
        for n_, p_ in m.named_parameters():
            assert rel_l2(p_.grad.cpu().numpy(), GR[pre + 'grad_' + n_]) < 2e-4, n_
    assert abs(m.diversity_penalty().item() - float(GR['sr_diversity_penalty'])) < 1e-6
    r = RNN(3 * nx * ny, hidden_dim=32)
    r.load_state_dict({k[len('rnn_param_'):]: T(GR[k], device='cpu') for k in GR.files if k.startswith('rnn_param_')}, strict=True)
    r = r.cuda()
    o, hdn = r(T(GR['rnn_in']))
    assert rel_l2(o.detach().cpu().numpy(), GR['rnn_out']) < 1e-5 and rel_l2(hdn.detach().cpu().numpy(), GR['rnn_hid']) < 1e-5
    ex = r.extrapolate(T(GR['rnn_in'])[:1, :1], 4)
    assert ex.shape == (1, 4, 3 * nx * ny) and rel_l2(ex.numpy(), GR['rnn_extrapolate']) < 1e-5
    o2, _ = r(torch.randn(3, nt, 3 * nx * ny, device='cuda'))          # batches > 1 work here (the reference's .view does not)
    assert o2.shape == (3, nt, 3 * nx * ny)


@pytest.mark.parametrize('shape', [
    # (T, K, C, P): every coefficient padding of the packed kernel (KMAX 4, 8, 10, 12, 16), ragged pixel tails, time
    # rows that do not fill a butterfly group, one and several time splits; and the generic kernel (K > 16, small P).
    # P % 4 == 0 and K <= 16 take the matrix-core gradient kernels, the others the packed-FMA ones (both stay covered)
    (37, 3, 3, 4096), (50, 7, 2, 5000), (64, 10, 3, 16384), (19, 10, 1, 4100), (41, 12, 3, 4096 + 1024 + 7),
    (33, 16, 2, 8192), (700, 10, 3, 4096), (2, 9, 3, 65536), (23, 20, 3, 4096), (29, 10, 3, 300),
    # the matrix-core gradient kernel (K <= 16, P % 4 == 0): smallest shapes, every K quad, a chunk boundary (T > 176) with a ragged tail
    (1, 1, 1, 4), (5, 2, 1, 8), (17, 16, 1, 64), (300, 13, 2, 260), (400, 5, 1, 1028)])
def test_basis_loss_kernels_direct(shape, gpu_device):
    """nns_basis_loss_fwd / _bwd / nns_basis_expand_bwd against float64 tensor contractions (rel-L2 2e-6: float32
    accumulation over up to 65536 pixels / 700 rows)."""
    from nns import ops
    T, K, C, P = shape
    rng = np.random.default_rng(T * 131 + K)
    coeff = rng.standard_normal((T, K, C)).astype(np.float32)
    basis = rng.standard_normal((K, C, P)).astype(np.float32)
    obs = rng.standard_normal((T, C, P)).astype(np.float32)
    c64, b64, o64 = coeff.astype(np.float64), basis.astype(np.float64), obs.astype(np.float64)
    pred = np.einsum('tkc,kcp->tcp', c64, b64)
    d = [torch.as_tensor(a, device='cuda') for a in (coeff, basis, obs)]
    ss = float(ops.basis_loss_fwd(*d).item())
    assert abs(ss - np.sum((pred - o64)**2)) < 1e-6 * np.sum((pred - o64)**2)
    scale = 0.37
    g = scale * (pred - o64)
    gc, gb = ops.basis_loss_bwd(*d, scale)
    assert rel_l2(gc.cpu().numpy(), np.einsum('tcp,kcp->tkc', g, b64)) < 2e-6
    assert rel_l2(gb.cpu().numpy(), np.einsum('tcp,tkc->kcp', g, c64)) < 2e-6
    ss3, gc3, gb3 = ops.basis_loss_fused(*d)                               # ONE sweep: loss and the unscaled gradient
    assert abs(float(ss3.item()) - np.sum((pred - o64)**2)) < 1e-6 * np.sum((pred - o64)**2)
    assert rel_l2(gc3.cpu().numpy(), np.einsum('tcp,kcp->tkc', pred - o64, b64)) < 2e-6
    assert rel_l2(gb3.cpu().numpy(), np.einsum('tcp,tkc->kcp', pred - o64, c64)) < 2e-6
    gc2, gb2 = ops.basis_expand_bwd(d[0], d[1], d[2])                      # upstream gradient = obs
    assert rel_l2(gc2.cpu().numpy(), np.einsum('tcp,kcp->tkc', o64, b64)) < 2e-6
    assert rel_l2(gb2.cpu().numpy(), np.einsum('tcp,tkc->kcp', o64, c64)) < 2e-6
    assert torch.allclose(ops.basis_expand(d[0], d[1]).cpu(), torch.as_tensor(pred, dtype=torch.float32), rtol=1e-4, atol=1e-4)


def test_cfg5_shape_loss_and_gradients_are_sums_over_chunks(gpu_device):
    """BASELINE config 5 at FULL shape -- T = nt * mb = 32 * 256 = 8192 time rows, K = 10, 3 channels, 256 x 256 pixels
    (6.4 GB of observations) -- through size-independent properties: loss^2 and the gradients of the whole batch (ONE
    fused sweep, nns_basis_loss_fused_f32) equal the sums over 8 chunks of 1024 rows, and chunk 0 is checked against the
    float64 contraction on the host restricted to a pixel sample (the full contraction is 2e11 flops of NumPy)."""
    from nns import ops
    T, K, C, n = 8192, 10, 3, 256
    P = n * n
    g = torch.Generator(device='cuda').manual_seed(3)
    coeff = torch.randn(T, K, C, device='cuda', generator=g) * 0.3
    basis = torch.randn(K, C, P, device='cuda', generator=g)
    obs = torch.randn(T, C, P, device='cuda', generator=g)
    ss, gc, gb = ops.basis_loss_fused(coeff, basis, obs)
    ss_fwd = ops.basis_loss_fwd(coeff, basis, obs)                      # the loss-only kernel agrees with the fused sweep
    assert abs(float(ss.item()) - float(ss_fwd.item())) < 1e-9 * float(ss.item())
    nchunk, rows = 8, T // 8
    ss_sum, gb_sum = 0.0, torch.zeros_like(gb, dtype=torch.float64)
    for c in range(nchunk):
        sl = slice(c * rows, (c + 1) * rows)
        s_c, gc_c, gb_c = ops.basis_loss_fused(coeff[sl].contiguous(), basis, obs[sl].contiguous())
        ss_sum += float(s_c.item())
        gb_sum += gb_c.double()
        assert rel_l2(gc[sl].cpu().numpy(), gc_c.cpu().numpy()) < 2e-6          # a row's coefficient gradient only sees its own row
        if c == 0:                                                      # float64 contraction on a sample of 4096 pixels
            idx = torch.arange(0, P, P // 4096, device='cuda')[:4096]
            c64, b64, o64 = coeff[sl].double(), basis[:, :, idx].double(), obs[sl][:, :, idx].double()
            r = torch.einsum('tkc,kcp->tcp', c64, b64) - o64
            ref_gb = torch.einsum('tcp,tkc->kcp', r, c64)
            assert rel_l2(gb_c[:, :, idx].cpu().numpy(), ref_gb.cpu().numpy()) < 2e-6
            s_small, gc_small, _ = ops.basis_loss_fused(coeff[sl].contiguous(), basis[:, :, idx].contiguous(), obs[sl][:, :, idx].contiguous())
            assert abs(float(s_small.item()) - float((r * r).sum().item())) < 1e-6 * float(s_small.item())
            assert rel_l2(gc_small.cpu().numpy(), torch.einsum('tcp,kcp->tkc', r, b64).cpu().numpy()) < 2e-6
    assert abs(ss_sum - float(ss.item())) < 1e-9 * ss_sum
    assert rel_l2(gb.cpu().numpy(), gb_sum.cpu().numpy()) < 2e-6


def test_ode_forward_row_kernel_matches_tile_kernel(tmp_path, gpu_device):
    """nns_ode_mlp_fwd_f32 has two kernels: one batch row per workgroup with the weights in registers (mb <= 4096, the default
    path of every test above) and the 16-row MFMA tile kernel (larger batches).  The tile kernel is forced in a child process
    (NNS_ODE_ROW_MAX=0 is read once per process) and must give the same trajectories to float32 rounding, for all three schemes
    and a ragged batch."""
    import os, subprocess, sys
    from conftest import PKG
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from nns import ops\n"
        "g = torch.Generator().manual_seed(11)\n"
        "K, mb, Nt = 30, 37, 9\n"
        "W = [torch.randn(128, K, generator=g) * 0.1, torch.zeros(128), torch.randn(128, 128, generator=g) * 0.1, torch.randn(128, generator=g) * 0.01,\n"
        "     torch.randn(K, 128, generator=g) * 0.1, torch.randn(K, generator=g) * 0.01]\n"
        "z0 = torch.randn(mb, K, generator=g)\n"
        "out = {m: ops.ode_mlp_fwd(z0.cuda(), *[w.cuda() for w in W], Nt, m).cpu().numpy() for m in ('Euler', 'RK2', 'RK4')}\n"
        "np.savez(sys.argv[1], **out)\n" % PKG)
    outs = {}
    for tag, env in (('row', {}), ('tile', {'NNS_ODE_ROW_MAX': '0'})):
        path = str(tmp_path / (tag + '.npz'))
        r = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(path)
    for m in ('Euler', 'RK2', 'RK4'):
        assert outs['row'][m].shape == (9, 37, 30)
        assert rel_l2(outs['row'][m], outs['tile'][m]) < 2e-6, m


@pytest.mark.parametrize('Nt,mb,K', [(100, 1, 30), (7, 3, 12), (41, 2, 33), (2, 1, 1), (1, 2, 5), (23, 5, 64), (300, 1, 9)])
def test_adjoint_chain_kernel_paths(Nt, mb, K, gpu_device):
    """nns_ode_adjoint_chain_f32 against a float64 loop: lam[Nt-1] = g[Nt-1], lam[s-1] = g[s-1] + lam[s] J[s] (anode/adjoint.py:38-70 as a
    recurrence on step Jacobians).  The cases walk its code paths: several LDS chunks (K = 30, Nt = 100: 19 steps per chunk), one chunk,
    K * K not a multiple of 4 (scalar copy loop), K > 32 (the 64-row instantiation), K = 1, a single time level, many chunks of tiny matrices."""
    from nns import ops
    g0 = torch.Generator().manual_seed(100 * Nt + K)
    J = (torch.randn(Nt, mb, K, K, generator=g0) * (0.5 / K ** 0.5) + torch.eye(K)).float()
    g = torch.randn(Nt, mb, K, generator=g0).float()
    lam = ops.ode_adjoint_chain(J.cuda(), g.cuda()).cpu().double()
    ref = torch.empty(Nt, mb, K, dtype=torch.float64)
    ref[Nt - 1] = g[Nt - 1].double()
    for s in range(Nt - 1, 0, -1):
        ref[s - 1] = g[s - 1].double() + torch.einsum('bi,bij->bj', ref[s], J[s].double())
    assert rel_l2(lam.numpy(), ref.numpy()) < 1e-5        # float32 recurrence: 3.8e-6 after 300 steps that grow to 1e12


def test_pixel_mlp_backward_exact_integers_many_supertiles(gpu_device):
    """The exact-integer indexing check on pixel counts where a workgroup of the fused backward walks SEVERAL 128-pixel super-tiles
    (the kernel launches at most one workgroup per CU: the small cases above give every workgroup at most one).  That is the path on which
    the next super-tile's inputs are requested during the current one's backward walk: a wrong prefetch pairing, a ragged last super-tile
    (pixel count not a multiple of 128) or pixels of one super-tile straddling two batch items show up as non-zero differences."""
    from nns import ops
    from oracle import neural as ON
    g = torch.Generator().manual_seed(23)
    for dims, shape in (([3, 64, 64, 3], (2, 200, 201)), ([2, 48, 3], (3, 171, 157)), ([3, 32, 32, 3], (5, 127, 113))):
        L = len(dims) - 1
        Ws = [((torch.rand(dims[i + 1], dims[i], generator=g) < 0.12).float() * (torch.randint(0, 2, (dims[i + 1], dims[i]), generator=g) * 2 - 1).float()) for i in range(L)]
        bs = [torch.randint(-1, 2, (dims[i + 1],), generator=g).float() for i in range(L)]
        x = torch.randint(-2, 3, (shape[0], dims[0]) + shape[1:], generator=g).float()
        gy = torch.randint(-1, 2, (shape[0], dims[-1]) + shape[1:], generator=g).float()
        assert shape[0] * shape[1] * shape[2] > 2 * 128 * 256 and (shape[0] * shape[1] * shape[2]) % 128 != 0
        ref_gx, ref_gW, ref_gb = ON.pixel_mlp_backward([w.double() for w in Ws], [b.double() for b in bs], x.double(), gy.double())
        h = x.double()
        for l in range(L):
            h = torch.einsum('oc,bcxy->boxy', Ws[l].double(), h) + bs[l].double()[None, :, None, None]
            assert h.abs().max() <= 256, (dims, l)
            h = torch.relu(h)
        for l in range(L):
            assert ref_gW[l].abs().max() < 2 ** 24 and ref_gb[l].abs().max() < 2 ** 24          # the float32 sums over all pixels stay exact
        gx, gW, gb = ops.pixel_mlp_bwd(x.cuda(), gy.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
        assert torch.equal(gx.cpu().double(), ref_gx), dims
        for l in range(L):
            assert torch.equal(gW[l].cpu().double(), ref_gW[l]), (dims, l)
            assert torch.equal(gb[l].cpu().double(), ref_gb[l]), (dims, l)


def test_pixel_mlp_backward_config3_full_shape_properties(gpu_device):
    """The fused bf16 backward at the shape the bench times it on (depth 8, width 64, 16 x 512^2 pixels; 128 super-tiles per workgroup)
    through size-independent properties: determinism (bitwise), batch independence of the input gradient (item b == the item alone,
    bitwise: a per-pixel operator), additivity of the parameter gradients over a split of the batch (float32 sums in a different order:
    1e-5), and the input gradient of a 4096-pixel sample against the float64 oracle with the kernel's bf16 operand rounding emulated."""
    from nns import ops
    from nns.neural_spectral.spectral_ode import PixelMLP
    from oracle import neural as ON
    torch.manual_seed(5)
    m = PixelMLP(8, 64).cuda()
    for b in m.biases:
        torch.nn.init.normal_(b, std=0.3)
    Ws, bs = [w.detach() for w in m.weights], [b.detach() for b in m.biases]
    x = torch.randn(16, 3, 512, 512, device='cuda')
    gy = torch.randn(16, 3, 512, 512, device='cuda')
    gx, gW, gb = ops.pixel_mlp_bwd(x, gy, Ws, bs)
    gx2, gW2, gb2 = ops.pixel_mlp_bwd(x, gy, Ws, bs)
    assert torch.equal(gx, gx2) and all(torch.equal(a, b) for a, b in zip(gW, gW2)) and all(torch.equal(a, b) for a, b in zip(gb, gb2))
    assert bool(torch.isfinite(gx).all())
    for b in (0, 7, 15):
        g1 = ops.pixel_mlp_bwd(x[b:b + 1].contiguous(), gy[b:b + 1].contiguous(), Ws, bs)[0]
        assert torch.equal(gx[b:b + 1], g1), b
    parts = [ops.pixel_mlp_bwd(x[q:q + 4].contiguous(), gy[q:q + 4].contiguous(), Ws, bs) for q in range(0, 16, 4)]
    for l in range(8):
        sW = sum(p[1][l].double() for p in parts); sb = sum(p[2][l].double() for p in parts)
        assert rel_l2(gW[l].cpu().numpy(), sW.cpu().numpy()) < 1e-5, l
        assert rel_l2(gb[l].cpu().numpy(), sb.cpu().numpy()) < 1e-5, l
    g = torch.Generator(device='cuda'); g.manual_seed(9)
    idx = torch.randint(0, 512 * 512, (4096,), device='cuda', generator=g)
    xs = x[11].reshape(3, -1)[:, idx].reshape(1, 3, 64, 64).contiguous()
    gs = gy[11].reshape(3, -1)[:, idx].reshape(1, 3, 64, 64).contiguous()
    ref_gx = ON.pixel_mlp_backward([w.cpu().double() for w in Ws], [b.cpu().double() for b in bs], xs.cpu().double(), gs.cpu().double(), bf16=True)[0]
    got = gx[11].reshape(3, -1)[:, idx].reshape(1, 3, 64, 64)
    assert rel_l2(got.cpu().numpy(), ref_gx.numpy()) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize('weight_decay,maximize', [(0.0, False), (0.01, False), (0.0, True)])
def test_adam_step_matches_torch_adam(weight_decay, maximize, gpu_device):
    """nns.optim.Adam (nns_adam_step_f32: ONE launch over all parameter tensors) against torch.optim.Adam -- the optimiser of the reference's
    training loops (spectral_ode.py:171,189) -- over six steps on tensors of awkward sizes (scalar tails, a 4-byte-aligned view, an empty
    tensor, more tensors than one launch table holds), and state_dict interchange in both directions."""
    import nns.optim as nns_optim
    torch.manual_seed(3)
    base = torch.randn(10007, device='cuda')
    shapes = [(1,), (3, 5), (128, 30), (2049,), (0,), (64, 64, 3)] + [(7,)] * 24
    a = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    base2 = base.clone()
    a.append(torch.nn.Parameter(base[1:4098])), b.append(torch.nn.Parameter(base2[1:4098]))       # contiguous, 4 bytes past a 16-byte boundary
    assert b[-1].data_ptr() % 16 == 4
    oa = torch.optim.Adam(a, lr=1e-2, weight_decay=weight_decay, maximize=maximize)
    ob = nns_optim.Adam(b, lr=1e-2, weight_decay=weight_decay, maximize=maximize)
    for step in range(6):
        gs = [torch.randn_like(p) * (10.0 ** (step - 3)) for p in a]
        for p, q, g in zip(a, b, gs):
            p.grad, q.grad = g.clone(), g.clone()
        oa.step(), ob.step()
        for p, q in zip(a, b):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (step, p.shape, (p - q).abs().max().item())
        if step == 2:                                               # swap the states through state_dict(): each class continues the other's run
            sa, sb = oa.state_dict(), ob.state_dict()
            oa.load_state_dict(sb), ob.load_state_dict(sa)
    for p, q in zip(a, b):
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa['step']) == float(sb['step']) == 6.0
        assert torch.allclose(sa['exp_avg'], sb['exp_avg'], rtol=2e-6, atol=1e-9) and torch.allclose(sa['exp_avg_sq'], sb['exp_avg_sq'], rtol=2e-6, atol=1e-12)


@pytest.mark.gpu
def test_graphed_backward_replays_the_training_iteration(gpu_device):
    """nns.graphs.GraphedBackward: loss + backward of the reference's training iteration (spectral_ode.py:178-188) captured ONCE as a HIP graph.
    Replays follow the parameters as the optimiser moves them: six graphed steps give the losses and parameters of six eager steps (float32
    atomics in the gradient sums: 1e-5), zero_grad(set_to_none=True) between steps is harmless, new observations go in through copy_."""
    import nns.optim as nns_optim
    from nns.graphs import GraphedBackward
    from nns.neural_spectral.spectral_ode import PDEFunc
    K, n, nt = 6, 64, 20
    torch.manual_seed(0)
    me = PDEFunc(K, n, n).cuda()
    mg = PDEFunc(K, n, n).cuda(); mg.load_state_dict(me.state_dict())
    obs = torch.randn(nt, 1, 3, n, n, device='cuda')
    obs_g = obs.clone()
    t = torch.arange(nt, device='cuda') + 1
    oe, og = nns_optim.Adam(me.parameters(), lr=1e-2), nns_optim.Adam(mg.parameters(), lr=1e-2)
    gb = GraphedBackward(mg.parameters(), lambda: mg.loss(obs_g[0], t, obs_g))
    for step in range(6):
        if step == 3:                                   # new data: into the tensors the graph reads
            obs = obs * 0.5 + 0.1
            obs_g.copy_(obs)
        oe.zero_grad()
        le = me.loss(obs[0], t, obs); le.backward(); oe.step()
        og.zero_grad(set_to_none=True)
        lg = gb(); og.step()
        assert abs(float(le) - float(lg)) <= 1e-5 * abs(float(le)), (step, float(le), float(lg))
    for a, b in zip(me.parameters(), mg.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), (a - b).abs().max().item()
