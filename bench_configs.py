#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (not the headline bench.py line): prints one JSON
object per config.  cfg1: chorin_fd 64x64 cavity step (float64, explicit + ADI); cfg2: neural_spectral 128x128
K=10 nt=100 training iteration + depth-4 pixel MLP; cfg3: 512x512 residual + depth-8 width-64 bf16 pixel MLP;
cfg5: ensemble 256 x 256x256 neural_spectral forward/backward (nt=32)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, _p)
import numpy as np
import torch


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def cfg1():
    from nns.chorin_fd import NavierStokesSystem
    from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
    n = 64
    dx = dy = 2. / (n - 1)
    u_bc = [D(0, 'left', dx, dy), D(1, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    v_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    p_bc = [D(0, 'top', dx, dy), N(0, 'bottom', dx, dy), N(0, 'left', dx, dy), N(0, 'right', dx, dy)]
    out = {}
    for method in ('explicit', 'semi_implicit'):
        for B in (1, 256):
            z = np.zeros((n, n)) if B == 1 else np.zeros((B, n, n))
            nt = 200                                                   # the reference driver's nt (:278)
            s = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=nt, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.02,
                                   beta=1.25, method=method)
            for graph in (True, False):
                dt = timeit(lambda: s.simulate_device(use_graph=graph), iters=2, warm=1) / nt
                tag = '%s_B%d_%s' % (method, B, 'graph' if graph else 'eager')
                out[tag + '_ms_per_step'] = 1e3 * dt
                out[tag + '_pt_steps_per_s'] = B * n * n / dt
    # the build's corrected options (section 8 (f) rank 3): true y-advection + red-black SOR
    for B in (1, 256):
        z = np.zeros((n, n)) if B == 1 else np.zeros((B, n, n))
        s = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=200, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.02,
                               beta=1.25, method='explicit', advection='corrected', pressure_solver='redblack')
        dt = timeit(lambda: s.simulate_device(use_graph=False), iters=2, warm=1) / 200
        out['explicit_corrected_redblack_B%d_eager_ms_per_step' % B] = 1e3 * dt
    return dict(config='cfg1 chorin_fd 64x64 cavity Re=100, nit=50, float64 (reference CPU: 0.51 s/step)', **out)


def cfg2():
    from nns.neural_spectral.spectral_ode import PDEFunc, PixelMLP
    K, n, nt = 10, 128, 100
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, 1, 3, n, n, device='cuda')
    t = torch.arange(nt, device='cuda') + 1
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def it():
        opt.zero_grad()
        m.loss(obs[0], t, obs).backward()
        opt.step()
    mlp = PixelMLP(4, 32).cuda()
    x = torch.randn(16, 3, n, n, device='cuda')
    tm = timeit(lambda: mlp(x), iters=20)
    return dict(config='cfg2 neural_spectral 128x128 K=10 nt=100 mb=1 float32 (reference CPU 51^2: fwd 1.49 s + bwd 2.44 s)',
                train_iter_ms=1e3 * timeit(it, iters=10), forward_ms=1e3 * timeit(lambda: m(obs[0], t), iters=10),
                pixel_mlp_d4_w32_fp32_Gpix_s=16 * n * n / tm / 1e9)


def cfg3():
    from nns.neural_spectral.spectral_ode import PixelMLP
    from nns.periodic import ResidualEngine
    from nns.synthetic import residual_inputs
    n, B = 512, 64
    f = [torch.as_tensor(np.tile(a, (B // 4, 1, 1)), device='cuda') for a in residual_inputs(4, n)]
    eng = ResidualEngine(n, n, 1e-3, 1.0, 2 * np.pi / 1000)
    o1 = tuple(torch.empty_like(f[0]) for _ in range(3)); o2 = tuple(torch.empty_like(f[0]) for _ in range(3))
    tr = timeit(lambda: eng.both(*f, out_fd=o1, out_spec=o2), iters=20)            # fused: spectral column pass + row pass with the stencil
    mlp = PixelMLP(8, 64).cuda()
    x = torch.randn(16, 3, n, n, device='cuda')
    out = {}
    for bf in (False, True):
        tm = timeit(lambda: mlp(x, bf16=bf), iters=10)
        flops = 2 * 16 * n * n * (3 * 64 + 6 * 64 * 64 + 64 * 3)
        out['pixel_mlp_d8_w64_%s' % ('bf16' if bf else 'fp32')] = dict(ms=1e3 * tm, Gpix_s=16 * n * n / tm / 1e9, TFLOPs=flops / tm / 1e12)
    # fused backward (bf16): forward recompute + data chain + weight gradients = 3x the forward's flops
    from nns import ops
    gy = torch.randn_like(x)
    ws, bs = [w.detach() for w in mlp.weights], [b.detach() for b in mlp.biases]
    tb = timeit(lambda: ops.pixel_mlp_bwd(x, gy, ws, bs), iters=10)
    out['pixel_mlp_d8_w64_bf16_backward'] = dict(ms=1e3 * tb, Gpix_s=16 * n * n / tb / 1e9, TFLOPs=3 * flops / tb / 1e12)
    # the physics-informed training step (data loss + residual of the prediction), Adam included
    from nns.neural_spectral.physics_informed import FieldStepper, train_step
    state = torch.as_tensor(np.stack([np.tile(a, (4, 1, 1)) for a in residual_inputs(4, n)[3:] + residual_inputs(4, n)[2:3]], axis=1), device='cuda')
    target = torch.as_tensor(np.stack([np.tile(a, (4, 1, 1)) for a in residual_inputs(4, n)[:3]], axis=1), device='cuda')
    for backend in ('fd9', 'spectral'):
        stepper = FieldStepper(8, 64).cuda()
        opt = torch.optim.Adam(stepper.parameters(), lr=1e-4)
        e2 = ResidualEngine(n, n, 1e-3, 1.0, 2 * np.pi / 1000, backend=backend)
        ts = timeit(lambda: train_step(stepper, e2, opt, state, target, lam=0.1), iters=10)
        out['physics_informed_step_d8_w64_bf16_%s' % backend] = dict(ms=1e3 * ts, Mpix_s=16 * n * n / ts / 1e6)
        s_cm, t_cm = state.transpose(0, 1).contiguous(), target.transpose(0, 1).contiguous()     # channel-major fields: no per-channel copies
        tc = timeit(lambda: train_step(stepper, e2, opt, s_cm, t_cm, lam=0.1, layout='cm'), iters=10)
        out['physics_informed_step_d8_w64_bf16_%s_channel_major' % backend] = dict(ms=1e3 * tc, Mpix_s=16 * n * n / tc / 1e6)
    return dict(config='cfg3 512x512 Re=1000: residual (FD5 + spectral, batch 64) + depth-8 width-64 pixel MLP',
                residual_updates_per_s=B * n * n / tr, **out)


def cfg5():
    from nns.neural_spectral.spectral_ode import PDEFunc
    K, n, nt, mb = 10, 256, 32, 256
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, mb, 3, n, n, device='cuda')                       # 6.4 GB
    t = torch.arange(nt, device='cuda') + 1

    def it():
        m.zero_grad()
        m.loss(obs[0], t, obs).backward()
    tm = timeit(it, iters=3, warm=1)
    return dict(config='cfg5 ensemble 256 x 256x256 neural_spectral, K=10, nt=32, float32 (single GPU: all 256 members)',
                fwd_bwd_ms=1e3 * tm, obs_GB=obs.numel() * 4 / 1e9, obs_stream_GBs=2 * obs.numel() * 4 / tm / 1e9)


if __name__ == '__main__':
    which = sys.argv[1:] or ['cfg1', 'cfg2', 'cfg3', 'cfg5']
    for w in which:
        print(json.dumps(globals()[w]()), flush=True)
