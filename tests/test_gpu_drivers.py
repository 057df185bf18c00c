"""The reference's driver scripts run unchanged against this package (PYTHONPATH = neural-navier-stokes_amd):
on-disk formats of SURVEY.md section 8 row f1 -- data_{method}.npz (u, v, p float64 [nt, nx, ny]),
checkpoint.pth.tar (model_state_dict, optimizer_state_dict, config, losses, penalties), extrapolation.npy."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, load_golden, rel_l2

pytestmark = pytest.mark.gpu


def run(script, args, cwd):
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, 'src', script)] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def test_chorin_fd_driver_then_neural_training_driver(tmp_path, gpu_device):
    d = str(tmp_path)
    run('chorin_fd/simulate.py', ['--nt', '5', '--nit', '50', '--nx', '64', '--ny', '64', '--nu', '0.02', '--method', 'explicit'], d)
    data = np.load(os.path.join(d, 'data_explicit.npz'))
    assert sorted(data.files) == ['p', 'u', 'v'] and data['u'].shape == (5, 64, 64) and data['u'].dtype == np.float64
    g = load_golden('chorin_fd_cavity_64_explicit_nu0.02.npz')                 # BASELINE config 1 through the driver
    assert rel_l2(data['u'][[0, -1]], g['u']) < 1e-9 and rel_l2(data['p'][[0, -1]], g['p']) < 1e-9
    run('direct_fd/simulate.py', ['--nt', '3', '--nx', '16', '--ny', '16'], d)
    assert np.load(os.path.join(d, 'data.npz'))['p'].shape == (3, 16, 16)
    for script in ('neural_spectral/spectral_ode.py', 'neural_spectral/spectral_ode2.py'):
        out = os.path.join(d, 'ck_' + os.path.basename(script)[:-3])
        r = run(script, ['--npz-path', os.path.join(d, 'data_explicit.npz'), '--out-dir', out, '--n-iters', '20', '--n-coeffs', '4'], d)
        assert 'graph capture' not in r.stdout                       # the iteration really was replayed from its HIP graph (a failed capture falls back and says so)
        ck = torch.load(os.path.join(out + '_4', 'checkpoint.pth.tar'), weights_only=False)
        assert sorted(ck.keys()) == ['config', 'losses', 'model_state_dict', 'optimizer_state_dict', 'penalties']
        assert ck['losses'].shape == (20,) and ck['losses'][-1] < ck['losses'][0]       # it trains
        assert ck['config'].n_coeffs == 4
        ex = np.load(os.path.join(out + '_4', 'extrapolation.npy'))
        assert ex.shape == (5, 3, 64, 64) and ex.dtype == np.float32


def test_physics_informed_training_step(gpu_device):
    """One-step field predictor trained on data + Navier-Stokes residual of its own prediction: every gradient comes
    from the fused HIP backward kernels (pixel MLP bf16, residual adjoint); the loss goes down and the MLP's
    parameter gradients match a pure-torch float32 replica of the same graph at bf16 tolerance."""
    import numpy as np
    import torch
    from nns.neural_spectral.physics_informed import FieldStepper, physics_informed_loss, train_step
    from nns.periodic import ResidualEngine
    from nns.synthetic import residual_inputs
    n, B, dt, nu = 64, 4, 1e-2, 0.05
    u, v, p, up, vp = residual_inputs(B, n)
    state = torch.as_tensor(np.stack([up, vp, p], axis=1), device='cuda')
    target = torch.as_tensor(np.stack([u, v, p], axis=1), device='cuda')
    eng = ResidualEngine(n, n, dt, 1.0, nu, backend='fd9')
    torch.manual_seed(0)
    model = FieldStepper(depth=4, width=32).cuda()
    # gradient check against torch autograd on an unfused float32 replica
    total, _, _ = physics_informed_loss(model, eng, state, target, lam=0.1)
    total.backward()
    def replica(x):
        h = x
        L = len(model.mlp.weights)
        for l in range(L):
            h = torch.einsum('oc,bcxy->boxy', model.mlp.weights[l], h) + model.mlp.biases[l][None, :, None, None]
            if l < L - 1:
                h = torch.relu(h)
        return x + h
    def res(uu, vv, pp, uo, vo):
        def d(f):
            fx = (torch.roll(f, -1, 1) - torch.roll(f, 1, 1)) / (2 * eng.dx); fy = (torch.roll(f, -1, 2) - torch.roll(f, 1, 2)) / (2 * eng.dy)
            dxx = torch.roll(f, -1, 1) - 2 * f + torch.roll(f, 1, 1); dyy = torch.roll(f, -1, 2) - 2 * f + torch.roll(f, 1, 2)
            xm, xp = torch.roll(f, 1, 1), torch.roll(f, -1, 1)
            corners = torch.roll(xm, 1, 2) + torch.roll(xm, -1, 2) + torch.roll(xp, 1, 2) + torch.roll(xp, -1, 2)
            cross = corners - 2 * (xm + xp + torch.roll(f, 1, 2) + torch.roll(f, -1, 2)) + 4 * f
            return fx, fy, dxx / eng.dx**2 + dyy / eng.dy**2 + (eng.dx**2 + eng.dy**2) / 12. * cross / (eng.dx**2 * eng.dy**2)
        ux, uy, lu = d(uu); vx, vy, lv = d(vv); px, py, _ = d(pp)
        return ((uu - uo) / dt + uu * ux + vv * uy + px - nu * lu, (vv - vo) / dt + uu * vx + vv * vy + py - nu * lv, ux + vy)
    ref_params = [q.detach().clone().requires_grad_(True) for q in model.parameters()]
    saved = [q.grad.clone() for q in model.parameters()]
    for q in model.parameters():
        q.grad = None
    pred = replica(state)
    r = res(pred[:, 0], pred[:, 1], pred[:, 2], state[:, 0], state[:, 1])
    ref_total = ((pred - target) ** 2).mean() + 0.1 * sum((x * x).mean() for x in r)
    ref_total.backward()
    assert abs(ref_total.item() - total.item()) < 3e-2 * abs(ref_total.item())
    for got, q in zip(saved, model.parameters()):
        num = (got - q.grad).norm().item(); den = q.grad.norm().item()
        assert num < 6e-2 * den + 1e-6, (num, den)
    # the fused loss head (default, round 4) against the same graph from tensor ops around ResidualEngine.differentiable
    for q in model.parameters():
        q.grad = None
    t_un, d_un, p_un = physics_informed_loss(model, eng, state, target, lam=0.1, fused=False)
    t_un.backward()
    assert abs(t_un.item() - total.item()) < 1e-5 * abs(total.item())
    for got, q in zip(saved, model.parameters()):
        assert (got - q.grad).norm().item() < 2e-3 * got.norm().item() + 1e-7
    # channel-major layout: the same loss and the same gradients (same kernels, other pixel order)
    for q in model.parameters():
        q.grad = None
    t_cm, d_cm, p_cm = physics_informed_loss(model, eng, state.transpose(0, 1).contiguous(), target.transpose(0, 1).contiguous(), lam=0.1, layout='cm')
    t_cm.backward()
    assert abs(t_cm.item() - total.item()) < 1e-5 * abs(total.item())
    for got, q in zip(saved, model.parameters()):
        assert (got - q.grad).norm().item() < 2e-3 * got.norm().item() + 1e-7
    # and it trains
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    hist = [train_step(model, eng, opt, state, target, lam=0.1)[0].item() for _ in range(40)]
    assert hist[-1] < 0.6 * hist[0], hist[::8]


def test_physics_informed_spectral_loss_on_reference_driver_grids(gpu_device):
    """The reference's drivers produce 51 x 51 fields (src/chorin_fd/simulate.py:280-281; 50 x 50 in src/direct_fd/simulate.py:153-154):
    the spectral physics loss must take them as they come.  Frames of the chorin_fd mirror's own cavity run go through
    ResidualEngine(backend='spectral') -- circulant-matrix form on both axes (csrc/spectral_dense.hip) -- forward and autograd backward
    against the float64 oracle, and the field stepper trains on them."""
    import numpy as np
    import torch
    from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
    from nns.chorin_fd import NavierStokesSystem
    from nns.neural_spectral.physics_informed import FieldStepper, train_step
    from nns.periodic import ResidualEngine
    from oracle import periodic as OP
    from conftest import rel_l2
    n, dt, nu, rho, Lbox = 51, 1e-3, 0.1, 1.0, 2.0
    dx = dy = 2. / (n - 1)
    u_bc = [D(0., 'left', dx, dy), D(1., 'right', dx, dy), D(0., 'top', dx, dy), D(0., 'bottom', dx, dy)]
    v_bc = [D(0., 'left', dx, dy), D(0., 'right', dx, dy), D(0., 'top', dx, dy), D(0., 'bottom', dx, dy)]
    p_bc = [D(0., 'top', dx, dy), N(0., 'bottom', dx, dy), N(0., 'left', dx, dy), N(0., 'right', dx, dy)]
    z = np.zeros((n, n))
    us, vs, ps = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=6, nit=50, nx=n, ny=n, dt=dt, rho=rho, nu=nu, beta=1.25,
                                    method='explicit').simulate()
    assert us.shape == (6, n, n) and np.abs(us[-1]).max() > 1e-3
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    u, v, p, up, vp = f32(us[1:]), f32(vs[1:]), f32(ps[1:]), f32(us[:-1]), f32(vs[:-1])          # 5 consecutive frame pairs
    eng = ResidualEngine(n, n, dt, rho, nu, Lbox, Lbox, backend='spectral')
    d = [torch.as_tensor(a, device='cuda').requires_grad_(True) for a in (u, v, p, up, vp)]
    r = eng.differentiable(*d)
    ref = OP.spectral_residual(*[a.astype(np.float64) for a in (u, v, p, up, vp)], dt, Lbox, Lbox, rho, nu)
    for g, w in zip(r, ref):
        assert rel_l2(g.detach().cpu().numpy(), w) <= 1e-6
    g3 = [torch.randn_like(t) for t in r]
    torch.autograd.backward(r, g3)
    refb = OP.spectral_residual_vjp(u.astype(np.float64), v.astype(np.float64), *[t.cpu().numpy().astype(np.float64) for t in g3], dt, Lbox, Lbox, rho, nu)
    for t, w in zip(d, refb):
        assert rel_l2(t.grad.cpu().numpy(), w) <= 1e-6
    # and the physics-informed step trains on these frames
    state = torch.as_tensor(np.stack([up, vp, p], axis=1), device='cuda')
    target = torch.as_tensor(np.stack([u, v, p], axis=1), device='cuda')
    torch.manual_seed(1)
    model = FieldStepper(depth=4, width=32).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    hist = [train_step(model, eng, opt, state, target, lam=1e-6)[0].item() for _ in range(30)]
    assert np.isfinite(hist).all() and hist[-1] < hist[0], hist[::6]


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['fd9_vec', 'fd5_wdiv', 'spectral_odd', 'spectral_pow2_channel_major', 'no_target'])
def test_physics_informed_loss_head_matches_the_oracle(case, gpu_device):
    """ops.PinnHeadFn (nns_pinn_assemble_f32 -> residual -> nns_pinn_loss_f32; adjoint -> nns_pinn_combine_f32) against oracle pinn_head in
    float64: the three losses and d total / d mlp_out, float4 and scalar (odd pixel count) paths, w_div != 1 (r_div scaled in place), no data
    term, channel-major fields; twice the same bits (the sums are deterministic); and the parts differentiated separately."""
    import numpy as np
    import torch
    from nns import ops
    from nns.periodic import ResidualEngine
    from oracle import periodic as OP
    from conftest import rel_l2
    B, n, backend, w_div, lam, with_target, cm = dict(
        fd9_vec=(3, 64, 'fd9', 1.0, 0.1, True, False), fd5_wdiv=(2, 48, 'fd5', 2.5, 0.3, True, False), spectral_odd=(5, 51, 'spectral', 1.5, 1e-3, True, False),
        spectral_pow2_channel_major=(4, 128, 'spectral', 1.0, 1e-2, True, True), no_target=(3, 37, 'fd9', 2.0, 1.0, False, False))[case]
    dt, rho, nu, L = 1e-2, 1.3, 0.05, 2.0
    rng = np.random.default_rng(11)
    out, state, target = (rng.standard_normal((B, 3, n, n)).astype(np.float32) * s for s in (0.1, 1.0, 1.0))
    eng = ResidualEngine(n, n, dt, rho, nu, L, L, backend=backend)
    kind, consts = eng.residual_spec()
    oc = consts if kind == 'fd' else consts[:5]
    want = OP.pinn_head(out.astype(np.float64), state.astype(np.float64), target.astype(np.float64) if with_target else None, kind, oc, lam, w_div)

    def dev(a):
        a = np.ascontiguousarray(a.transpose(1, 0, 2, 3)).reshape(1, 3, -1) if cm else a
        return torch.as_tensor(a, device='cuda')

    def run():
        o = dev(out).requires_grad_(True)
        t3 = ops.PinnHeadFn.apply(o, dev(state), dev(target) if with_target else None, (kind, consts), (B, n, n), lam, w_div)
        t3[0].backward()
        g = o.grad.cpu().numpy()
        g = g.reshape(3, B, n, n).transpose(1, 0, 2, 3) if cm else g
        return [float(x.detach()) for x in t3], g, o

    got, grad, _ = run()
    for a, b in zip(got, want[:3]):
        assert abs(a - b) <= 2e-5 * abs(b) + 1e-12, (got, want[:3])
    assert rel_l2(grad, want[3]) <= 2e-5, rel_l2(grad, want[3])
    got2, grad2, _ = run()
    assert got2 == got and np.array_equal(grad, grad2)
    # d (3 data + 5 phys) / d out through the separate outputs = the oracle's gradient with the two terms reweighted
    o = dev(out).requires_grad_(True)
    t3 = ops.PinnHeadFn.apply(o, dev(state), dev(target) if with_target else None, (kind, consts), (B, n, n), lam, w_div)
    (3.0 * t3[1] + 5.0 * t3[2]).backward()
    g_phys = OP.pinn_head(out.astype(np.float64), state.astype(np.float64), None, kind, oc, 1.0, w_div)[3]
    g_data = (want[3] - lam * g_phys) if with_target else 0.0
    g = o.grad.cpu().numpy()
    g = g.reshape(3, B, n, n).transpose(1, 0, 2, 3) if cm else g
    assert rel_l2(g, 3.0 * g_data + 5.0 * g_phys) <= 2e-5
