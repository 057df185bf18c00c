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


def timeit3(fn, iters=10, warm=2):
    """Seconds per call as the MEDIAN of three back-to-back blocks of `iters` calls (returns (median, the three blocks)).  Round 4: the pool's
    boxes show a sporadic device-side stall of 40-70 ms (seen in the middle of un-profiled kernel streams, with round 3's library as with this
    one: tools/c5_stall_probe2.py); one of them inside a single block of four 2 ms steps read as a 22 ms step in a driver-style run.  The
    median of three blocks drops one such outlier without turning the figure into a best-of."""
    blocks = [timeit(fn, iters=iters, warm=warm if r == 0 else 0) for r in range(3)]
    return sorted(blocks)[1], blocks


def _cpu_time(fn, budget_s=3.0, max_reps=50):
    """Median seconds of fn() on this host (time.perf_counter), after one warm-up, within ~budget_s."""
    fn()
    ts, t_all = [], time.perf_counter()
    while len(ts) < max_reps and (len(ts) < 3 or time.perf_counter() - t_all < budget_s):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def cpu_baseline_operators(n=1024):
    """the reference's own operators -- as the NumPy float64 oracle port (oracle/chorin_fd.py,
    oracle/direct_fd.py; bitwise the reference on the goldens) -- timed on THIS box's host, one thread (element-wise NumPy):
    explicit predictor (src/chorin_fd/simulate.py:63-91), pressure RHS (:186-188), correction (:204-210), _build_up_b
    (src/direct_fd/simulate.py:56-66), one Jacobi sweep (:76-86) at n x n; the lexicographic SOR loop (:190-196) at 64 x 64."""
    from oracle import chorin_fd as OC, direct_fd as OD
    rng = np.random.default_rng(0)
    h = 2. / (n - 1)
    u, v, u1, v1, p = (rng.standard_normal((n, n)) for _ in range(5))
    rows = {}

    def add(name, fn, pts, ref):
        t, reps = _cpu_time(fn)
        rows[name] = dict(ms=1e3 * t, Mpt_s=pts / t / 1e6, reps=reps, reference=ref)
    add('explicit_predictor_%d' % n, lambda: OC.explicit_predictor(u, v, u1, v1, 1e-3, h, h, 0.02), n * n, 'src/chorin_fd/simulate.py:63-91')
    add('pressure_rhs_%d' % n, lambda: OC.pressure_rhs(u, v, 1e-3, h, h, 1.0), n * n, 'src/chorin_fd/simulate.py:186-188')
    add('correction_%d' % n, lambda: OC.correction(u, v, p, 1e-3, h, h), n * n, 'src/chorin_fd/simulate.py:204-210')
    add('build_up_b_%d' % n, lambda: OD.build_up_b(u, v, 1e-3, h, h, 1.0), n * n, 'src/direct_fd/simulate.py:56-66')
    b = OD.build_up_b(u, v, 1e-3, h, h, 1.0)
    add('jacobi_sweep_%d' % n, lambda: OD.jacobi_sweep(p, b, h, h), n * n, 'src/direct_fd/simulate.py:76-86')
    m = 64
    hm = 2. / (m - 1)
    pm, Cm = 0.01 * rng.standard_normal((m, m)), 0.1 * rng.standard_normal((m, m))
    add('sor_sweep_lexicographic_64', lambda: OC.sor_sweep_lexicographic(pm, Cm, hm, hm, 1.25), m * m, 'src/chorin_fd/simulate.py:190-196 (pure-Python loop)')
    return dict(value=rows['explicit_predictor_%d' % n]['Mpt_s'] * 1e6, unit='grid points/s (explicit predictor, %d^2)' % n, cores=1, kind='port',
                sample='NumPy float64 oracle port on this host, median of >= 3 runs within ~3 s per operator', host_cores_present=os.cpu_count(), operators=rows)


def cpu_baseline_cfg1(n=64, nit=50, steps=2):
    """One chorin_fd cavity step of the oracle port on this host (the reference's CPU path: ~95 % pure-Python SOR loop)."""
    from oracle import chorin_fd as OC
    from oracle.boundary import cavity_bcs
    h = 2. / (n - 1)
    u_bc, v_bc, p_bc = cavity_bcs(h, h)
    z = np.zeros((n, n))
    t0 = time.perf_counter()
    OC.simulate(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, steps, nit, 1e-3, 1.0, 0.02, 1.25, 'explicit')
    t = (time.perf_counter() - t0) / steps
    return dict(value=n * n / t, unit='grid-point steps/s', cores=1, kind='port', ms_per_step=1e3 * t,
                sample='%d explicit steps of the %dx%d cavity, nit=%d, NumPy float64 oracle port (src/chorin_fd/simulate.py:212-234)' % (steps, n, n, nit),
                host_cores_present=os.cpu_count())


def cpu_baseline_neural(K=10, n=51, nt=100, budget_s=6.0):
    """spectral_ode.PDEFunc forward + backward on the host CPU with torch (float32, the reference's own shapes: 51 x 51, nt = 100,
    mb = 1; src/neural_spectral/spectral_ode.py:62-81,178-190) through the oracle's restatement."""
    from oracle import neural as ON
    torch.manual_seed(0)
    init = torch.randn(3 * K, requires_grad=True)
    mlp = [torch.randn(128, 3 * K) * 0.1, torch.zeros(128), torch.randn(128, 128) * 0.1, torch.zeros(128), torch.randn(3 * K, 128) * 0.1, torch.zeros(3 * K)]
    for t in mlp:
        t.requires_grad_(True)
    basis = torch.randn(K, 3, n, n, requires_grad=True)
    obs = torch.randn(nt, 1, 3, n, n)
    nth = torch.get_num_threads()

    def fwd_bwd():
        pred, _ = ON.pde_forward(init, mlp, basis, 1, nt)
        ON.loss_fn(pred, obs).backward()
    t, reps = _cpu_time(fwd_bwd, budget_s=budget_s, max_reps=5)
    return dict(value=1.0 / t, unit='training iterations/s (forward + backward)', cores=nth, kind='port', ms_per_iter=1e3 * t,
                sample='%d x forward+backward of PDEFunc K=%d, %dx%d, nt=%d, mb=1, torch CPU float32 (%d threads)' % (reps, K, n, n, nt, nth),
                host_cores_present=os.cpu_count())


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
    return dict(config='cfg1 chorin_fd 64x64 cavity Re=100, nit=50, float64', cpu_baseline=cpu_baseline_cfg1(), **out)


def cfg2():
    from nns.neural_spectral.spectral_ode import PDEFunc, PixelMLP
    K, n, nt = 10, 128, 100
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, 1, 3, n, n, device='cuda')
    t = torch.arange(nt, device='cuda') + 1
    import nns.optim as nns_optim
    opt = nns_optim.Adam(m.parameters(), lr=1e-3)

    def it():
        opt.zero_grad()
        m.loss(obs[0], t, obs).backward()
        opt.step()
    mlp = PixelMLP(4, 32).cuda()
    x = torch.randn(16, 3, n, n, device='cuda')
    tm = timeit(lambda: mlp(x), iters=20)
    return dict(config='cfg2 neural_spectral 128x128 K=10 nt=100 mb=1 float32', cpu_baseline=cpu_baseline_neural(),
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
        import nns.optim as nns_optim
        opt = nns_optim.Adam(stepper.parameters(), lr=1e-4)
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
                fwd_bwd_ms=1e3 * tm, obs_GB=obs.numel() * 4 / 1e9, obs_stream_GBs=obs.numel() * 4 / tm / 1e9, note="observations are read ONCE per step (fused loss + gradient sweep)")


HBM_PEAK_GBS, BF16_PEAK_TF, F32_PEAK_TF = 8000.0, 2500.0, 157.3          # MI355X_MICROARCH.md: HBM3E spec, dense bf16 MFMA, f32 vector = f32 MFMA


def secondary(cpu=True):
    """The other BASELINE.json configs under the same clock as the headline (bench.py appends this object to its JSON line; ~20 s):
    cfg 1 step, cfg 2 training iteration, cfg 3 residual at 512^2 + depth-8 width-64 MLP forward / backward, cfg 5 step -- each with
    the roofline that bounds it (`bound`, `achieved`, `peak`, `frac`, algorithmic bytes or FLOPs) and, where the reference has a CPU
    path for it, the oracle port timed on this box's host (`cpu_baseline`, bounded samples)."""
    from nns import ops
    out = {}
    # ---- cfg 1: chorin_fd 64 x 64 cavity, Re = 100, nit = 50, float64, explicit predictor (src/chorin_fd/simulate.py:212-234)
    from nns.chorin_fd import NavierStokesSystem
    from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
    n = 64
    dx = dy = 2. / (n - 1)
    u_bc = [D(0, 'left', dx, dy), D(1, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    v_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    p_bc = [D(0, 'top', dx, dy), N(0, 'bottom', dx, dy), N(0, 'left', dx, dy), N(0, 'right', dx, dy)]
    z = np.zeros((n, n))
    nt = 100
    sysm = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=nt, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.02, beta=1.25, method='explicit')
    t1 = timeit(lambda: sysm.simulate_device(use_graph=False), iters=2, warm=1) / nt
    S = 49                                                                  # at most nit - 1 sweeps (:183,:190)
    b1 = (56 + 12 * S) * 2.0 * n * n                                        # 56 + 12 S B/pt in float32, x 2 for float64
    sys_si = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=nt, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.02, beta=1.25, method='semi_implicit')
    t1si = timeit(lambda: sys_si.simulate_device(use_graph=False), iters=2, warm=1) / nt
    out['cfg1_chorin_fd_64_step'] = dict(ms=1e3 * t1, semi_implicit_ms=1e3 * t1si, method='explicit (the line BASELINE config 1 is timed on); semi_implicit_ms = the reference driver\'s default method',
                                         grid_point_steps_per_s=n * n / t1, dtype='f64', bound='latency (sequential SOR fronts of ONE 64^2 grid; HBM row for scale)',
                                         algorithmic_bytes=b1, achieved=b1 / t1 / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=b1 / t1 / 1e9 / HBM_PEAK_GBS,
                                         cpu_baseline=cpu_baseline_cfg1() if cpu else None)
    # ---- cfg 2: neural_spectral 128 x 128, K = 10, nt = 100, mb = 1: one training iteration (src/neural_spectral/spectral_ode.py:178-190)
    from nns.neural_spectral.spectral_ode import PDEFunc, PixelMLP
    K, n, nt = 10, 128, 100
    torch.manual_seed(0)
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, 1, 3, n, n, device='cuda')
    t = torch.arange(nt, device='cuda') + 1
    import nns.optim as nns_optim
    opt = nns_optim.Adam(m.parameters(), lr=1e-3)

    def it2():
        opt.zero_grad()
        m.loss(obs[0], t, obs).backward()
        opt.step()
    t2e, reps2e = timeit3(it2, iters=10, warm=3)   # eager: ~35 small launches per iteration, host jitter moves a single block of 10 by +-30 %
    # what nns/neural_spectral/train.py runs: loss + backward captured once as a HIP graph (nns/graphs.py), replayed, then the optimiser step
    graph_note = 'loss + backward replayed from ONE HIP graph (nns.graphs.GraphedBackward, the default of train.py) + the optimiser step'
    try:
        from nns.graphs import GraphedBackward
        gb2 = GraphedBackward(m.parameters(), lambda: m.loss(obs[0], t, obs))

        def it2g():
            gb2()
            opt.step()
        t2, reps2 = timeit3(it2g, iters=10, warm=3)
    except Exception as e:                         # noqa: BLE001 -- the eager figure stands in, and the line says why
        t2, reps2, graph_note = t2e, reps2e, 'eager (graph capture failed: %r)' % (e,)
    b2 = obs.numel() * 4.0
    f2 = 3 * 4 * nt * 2.0 * (30 * 128 + 128 * 128 + 128 * 30)              # RK4, nt steps, forward + 2x backward, one trajectory
    out['cfg2_neural_spectral_128_train_iter'] = dict(ms=1e3 * t2, timing='median of 3 blocks of 10 iterations', how=graph_note, ms_blocks=[1e3 * r for r in reps2],
                                                      eager_ms=1e3 * t2e, eager_ms_blocks=[1e3 * r for r in reps2e], iterations_per_s=1.0 / t2, dtype='f32', bound='latency (nt = 100 dependent RK4 steps of a 30-128-128-30 MLP; HBM row for scale)',
                                                      algorithmic_bytes=b2, ode_flops=f2, achieved=b2 / t2 / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=b2 / t2 / 1e9 / HBM_PEAK_GBS,
                                                      cpu_baseline=cpu_baseline_neural(budget_s=3.0) if cpu else None)
    gb2 = None
    del m, obs, opt
    # ---- cfg 3: 512 x 512, Re = 1000: residual (FD 5-point + spectral) of 64 grids; depth-8 width-64 MLP on 16 x 512^2 pixels, bf16 on MFMA
    from nns.periodic import ResidualEngine
    from nns.synthetic import residual_inputs
    n, B = 512, 64
    f = [torch.as_tensor(np.tile(a, (B // 4, 1, 1)), device='cuda') for a in residual_inputs(4, n)]
    eng = ResidualEngine(n, n, 1e-3, 1.0, 2 * np.pi / 1000)
    o1 = tuple(torch.empty_like(f[0]) for _ in range(3)); o2 = tuple(torch.empty_like(f[0]) for _ in range(3))
    t3 = timeit3(lambda: eng.both(*f, out_fd=o1, out_spec=o2), iters=20, warm=5)[0]
    b3 = 80.0 * B * n * n
    out['cfg3_residual_512'] = dict(ms=1e3 * t3, residual_updates_per_s=B * n * n / t3, dtype='f32', bound='hbm', bytes_per_pt_two_pass=80.0, algorithmic_bytes=b3,
                                    achieved=b3 / t3 / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=b3 / t3 / 1e9 / HBM_PEAK_GBS)
    del f, o1, o2
    mlp = PixelMLP(8, 64).cuda()
    x = torch.randn(16, 3, n, n, device='cuda')
    flops = 2.0 * 16 * n * n * (3 * 64 + 6 * 64 * 64 + 64 * 3)
    with torch.no_grad():
        tf = timeit3(lambda: mlp(x, bf16=True), iters=10, warm=3)[0]
        tf32 = timeit3(lambda: mlp(x, bf16=False), iters=4, warm=2)[0]
    gy = torch.randn_like(x)
    ws, bs = [w.detach() for w in mlp.weights], [b.detach() for b in mlp.biases]
    tb = timeit3(lambda: ops.pixel_mlp_bwd(x, gy, ws, bs), iters=10, warm=3)[0]
    out['cfg3_mlp_d8_w64_forward_bf16'] = dict(ms=1e3 * tf, dtype='bf16 (f32 accumulate)', bound='mfma', useful_flops=flops, achieved=flops / tf / 1e12, peak=BF16_PEAK_TF,
                                              unit='TFLOP/s', frac=flops / tf / 1e12 / BF16_PEAK_TF)
    out['cfg3_mlp_d8_w64_forward_f32'] = dict(ms=1e3 * tf32, dtype='f32', bound='mfma', useful_flops=flops, achieved=flops / tf32 / 1e12, peak=F32_PEAK_TF, unit='TFLOP/s',
                                             frac=flops / tf32 / 1e12 / F32_PEAK_TF)
    out['cfg3_mlp_d8_w64_backward_bf16'] = dict(ms=1e3 * tb, dtype='bf16 (f32 accumulate)', bound='mfma', useful_flops=3 * flops, achieved=3 * flops / tb / 1e12, peak=BF16_PEAK_TF,
                                               unit='TFLOP/s', frac=3 * flops / tb / 1e12 / BF16_PEAK_TF,
                                               note='forward recompute + data chain + weight gradients = 3x the forward FLOPs, one launch, no saved activations')
    del mlp, x, gy
    # ---- cfg 5: ensemble of 256 initial conditions x 256^2, K = 10, nt = 32: loss + backward of one step on ONE GPU (all 256 members)
    K, n, nt, mb = 10, 256, 32, 256
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, mb, 3, n, n, device='cuda')                       # 6.4 GB
    t = torch.arange(nt, device='cuda') + 1

    def it5():
        m.zero_grad()
        m.loss(obs[0], t, obs).backward()
    t5, reps5 = timeit3(it5, iters=4, warm=2)
    b5 = obs.numel() * 4.0
    out['cfg5_ensemble_256x256_step'] = dict(ms=1e3 * t5, timing='median of 3 blocks of 4 steps', ms_blocks=[1e3 * r for r in reps5], dtype='f32', bound='hbm', algorithmic_bytes=b5, achieved=b5 / t5 / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                                             frac=b5 / t5 / 1e9 / HBM_PEAK_GBS, note='forward + backward; the 6.4 GB of observations cross HBM once per step (fused loss + gradient sweep)')
    del m, obs
    torch.cuda.empty_cache()
    return out


def cpu_ops():
    return dict(config='reference operators at 1024^2 on this host (BASELINE.md section 3)', cpu_baseline=cpu_baseline_operators())


if __name__ == '__main__':
    which = sys.argv[1:] or ['cpu_ops', 'cfg1', 'cfg2', 'cfg3', 'cfg5']
    for w in which:
        print(json.dumps(globals()[w]()), flush=True)
