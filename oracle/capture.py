#!/usr/bin/env python3
"""Golden-vector capture (test infrastructure only; runs ONLY in the build container).

Imports the reference (``/root/reference``, read-only) in fresh subprocesses, feeds it seeded
inputs and freezes inputs + outputs as small ``.npz`` fixtures under ``tests/golden/``.  The
fixtures are DATA (arrays + scalars); no reference source, bytecode or pickled callables are
stored, and nothing at test time reads ``/root/reference``.

    python oracle/capture.py            # regenerate every fixture

Accommodations (none changes reference arithmetic):
  1. ``torchdiffeq`` is not installed and is imported-but-unused by the reference
     (src/neural_spectral/spectral_ode.py:10): an empty stand-in module is placed in
     ``sys.modules`` before the import.
  2. ``semi_implicit`` builds a ragged ``np.array([...])`` (src/chorin_fd/simulate.py:105-121),
     an error on NumPy >= 1.24: the module-global ``np`` of src.chorin_fd.simulate is replaced by
     a proxy whose ``array`` falls back to ``dtype=object`` (what old NumPy did silently); the
     same proxy counts ``np.max`` calls so the SOR sweep count / final err can be recorded.
  3. each reference module is imported in its own subprocess, because chorin_fd and
     chorin_spectral install a process-global ``warnings.filterwarnings('error')``.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference'


def smooth_field(rng, nx, ny, amp=1.0):
    """Smooth field + 0.1 * white noise, float64, O(amp)."""
    x = np.linspace(-1, 1, nx)[:, None]
    y = np.linspace(-1, 1, ny)[None, :]
    a = rng.uniform(0.5, 1.5, size=4)
    ph = rng.uniform(0, 2 * np.pi, size=4)
    f = (np.sin(a[0] * np.pi * x + ph[0]) * np.cos(a[1] * np.pi * y + ph[1]) +
         0.5 * np.cos(2 * a[2] * np.pi * x + ph[2]) * np.sin(a[3] * np.pi * y + ph[3]))
    return amp * (f + 0.1 * rng.standard_normal((nx, ny)))


def bc_tuples(bcs):
    return [(b.type, b.boundary, float(b.value), float(b.dx), float(b.dy)) for b in bcs]


def pack_bcs(prefix, bcs, out):
    """BC list -> arrays (kind id, side id, value, dx, dy) so that the fixture stays pure data."""
    kinds = {'dirichlet': 0, 'neumann': 1}
    sides = {'left': 0, 'right': 1, 'bottom': 2, 'top': 3}
    out[prefix + '_kind'] = np.array([kinds[b[0]] for b in bcs], dtype=np.int64)
    out[prefix + '_side'] = np.array([sides[b[1]] for b in bcs], dtype=np.int64)
    out[prefix + '_value'] = np.array([b[2] for b in bcs], dtype=np.float64)
    out[prefix + '_dx'] = np.array([b[3] for b in bcs], dtype=np.float64)
    out[prefix + '_dy'] = np.array([b[4] for b in bcs], dtype=np.float64)


def cavity_bcs(B, dx, dy):
    D, Nm = B.DirichletBoundaryCondition, B.NeumannBoundaryCondition
    u_bc = [D(0, 'left', dx, dy), D(1, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    v_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    p_bc = [D(0, 'top', dx, dy), Nm(0, 'bottom', dx, dy), Nm(0, 'left', dx, dy), Nm(0, 'right', dx, dy)]
    return u_bc, v_bc, p_bc


# ----------------------------------------------------------------------------- workers
def worker_boundary():
    import src.boundary as B
    rng = np.random.default_rng(0)
    out = {}
    A0 = smooth_field(rng, 8, 6)
    out['A0'] = A0
    dx, dy = 0.25, 0.4
    for kind, cls in (('dirichlet', B.DirichletBoundaryCondition), ('neumann', B.NeumannBoundaryCondition)):
        for side in ('left', 'right', 'bottom', 'top'):
            A = A0.copy()
            r = cls(0.7, side, dx, dy).apply(A)
            assert r is A
            out['{}_{}'.format(kind, side)] = A
    out['single_value'] = np.float64(0.7)
    out['single_dx'], out['single_dy'] = np.float64(dx), np.float64(dy)
    S0 = smooth_field(rng, 8, 8)
    out['S0'] = S0
    d = 2. / 7.
    for name, bcs in zip(('u', 'v', 'p'), cavity_bcs(B, d, d)):
        A = S0.copy()
        for bc in bcs:
            A = bc.apply(A)
        out['list_' + name] = A
        pack_bcs('list_' + name + '_bc', bc_tuples(bcs), out)
    # a non-trivial mixed list (values != 0) for corner-order coverage
    mixed = [B.NeumannBoundaryCondition(0.3, 'left', d, d), B.DirichletBoundaryCondition(-1.5, 'top', d, d),
             B.NeumannBoundaryCondition(-0.8, 'top', d, d), B.NeumannBoundaryCondition(0.25, 'right', d, d),
             B.DirichletBoundaryCondition(2.0, 'bottom', d, d), B.NeumannBoundaryCondition(1.1, 'bottom', d, d)]
    A = S0.copy()
    for bc in mixed:
        A = bc.apply(A)
    out['list_mixed'] = A
    pack_bcs('list_mixed_bc', bc_tuples(mixed), out)
    np.savez_compressed(os.path.join(GOLD, 'boundary.npz'), **out)


class _NpProxy(object):
    """Stand-in for the module-global ``np`` of src.chorin_fd.simulate (accommodation 2)."""

    def __init__(self):
        self.max_calls = 0
        self.last_max = None

    def __getattr__(self, name):
        return getattr(np, name)

    def array(self, obj, *a, **k):
        try:
            return np.array(obj, *a, **k)
        except ValueError:
            return np.array(obj, dtype=object)

    def max(self, *a, **k):
        self.max_calls += 1
        self.last_max = np.max(*a, **k)
        return self.last_max


def worker_chorin_fd():
    import src.boundary as B
    import src.chorin_fd.simulate as M
    proxy = _NpProxy()
    M.np = proxy
    M.tqdm = lambda x, *a, **k: x
    rng = np.random.default_rng(0)
    for n in (16, 64):
        out = {}
        dx = dy = 2. / (n - 1)
        u_bc, v_bc, p_bc = cavity_bcs(B, dx, dy)
        dt, rho, nu, beta = 1e-3, 1.0, 0.1, 1.25
        out['params'] = np.array([dt, rho, nu, beta, dx, dy])
        for nm, bcs in zip('uvp', (u_bc, v_bc, p_bc)):
            pack_bcs(nm + '_bc', bc_tuples(bcs), out)
        u, v, u1, v1 = [smooth_field(rng, n, n) for _ in range(4)]
        p0 = smooth_field(rng, n, n, amp=0.3)
        out.update(u=u, v=v, u1=u1, v1=v1, p0=p0)
        for method in ('explicit', 'semi_implicit'):
            s = M.NavierStokesSystem(None, None, None, u_bc, v_bc, p_bc, nt=1, nit=50, nx=n, ny=n,
                                     dt=dt, rho=rho, nu=nu, beta=beta, method=method)
            f = s._explicit_predictor_step if method == 'explicit' else s._semi_implicit_predictor_step
            ui, vi = f(u.copy(), v.copy(), u1.copy(), v1.copy())
            out['pred_%s_ui' % method], out['pred_%s_vi' % method] = ui, vi
        s = M.NavierStokesSystem(None, None, None, u_bc, v_bc, p_bc, nt=1, nit=50, nx=n, ny=n,
                                 dt=dt, rho=rho, nu=nu, beta=beta, method='explicit')
        # pressure solve on small-divergence inputs (so the tolerance path is also exercised)
        ui_s, vi_s = 1e-4 * u, 1e-4 * v
        out['press_ui'], out['press_vi'] = ui_s, vi_s
        for nit in (3, 50, 400):
            s.nit = nit
            proxy.max_calls = 0
            p = p0.copy() * (1e-3 if nit == 400 else 1.0)
            out['press_p0_nit%d' % nit] = p.copy()
            r = s._get_pressure(ui_s.copy(), vi_s.copy(), p)
            assert r is p
            out['press_p_nit%d' % nit] = p
            out['press_sweeps_nit%d' % nit] = np.int64(proxy.max_calls)
            out['press_err_nit%d' % nit] = np.float64(proxy.last_max)
        un1, vn1 = s._correction_step(u.copy(), v.copy(), p0.copy())
        out['corr_u'], out['corr_v'] = un1, vn1
        for method in ('explicit', 'semi_implicit'):
            s = M.NavierStokesSystem(None, None, None, u_bc, v_bc, p_bc, nt=1, nit=20, nx=n, ny=n,
                                     dt=dt, rho=rho, nu=nu, beta=beta, method=method)
            proxy.max_calls = 0
            a, b, c = s.step(0.1 * u, 0.1 * v, 0.1 * u1, 0.1 * v1, 0.01 * p0.copy())
            out['step_%s_u' % method], out['step_%s_v' % method], out['step_%s_p' % method] = a, b, c
            out['step_%s_sweeps' % method] = np.int64(proxy.max_calls)
        np.savez_compressed(os.path.join(GOLD, 'chorin_fd_ops_%d.npz' % n), **out)

    # cavity trajectories: cfg 1 of BASELINE.json = 64x64, Re = 100 (nu = U L / Re, L = 2 -> 0.02;
    # L = 1 convention -> 0.01), dt = 1e-3, rho = 1, beta = 1.25, nit = 50, lid U = 1 on 'right'.
    for n, nt, nus in ((16, 10, (0.1,)), (64, 5, (0.02, 0.01))):
        for method in ('explicit', 'semi_implicit'):
            for nu in nus:
                dx = dy = 2. / (n - 1)
                u_bc, v_bc, p_bc = cavity_bcs(B, dx, dy)
                z = np.zeros((n, n))
                s = M.NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=nt, nit=50,
                                         nx=n, ny=n, dt=1e-3, rho=1, nu=nu, beta=1.25, method=method)
                ul, vl, pl = s.simulate()
                out = dict(params=np.array([1e-3, 1.0, nu, 1.25, dx, dy]), nt=np.int64(nt), nit=np.int64(50))
                if n == 16:
                    out.update(u=ul, v=vl, p=pl)
                else:   # keep the fixture small: first and last step only
                    out.update(u=ul[[0, -1]], v=vl[[0, -1]], p=pl[[0, -1]])
                np.savez_compressed(os.path.join(GOLD, 'chorin_fd_cavity_%d_%s_nu%g.npz' % (n, method, nu)), **out)


def worker_direct_fd():
    import src.boundary as B
    import src.direct_fd.simulate as M
    M.tqdm = lambda x, *a, **k: x
    rng = np.random.default_rng(0)
    for n in (16, 64):
        out = {}
        dx = dy = 2. / (n - 1)
        u_bc, v_bc, p_bc = cavity_bcs(B, dx, dy)
        dt, rho, nu = 1e-3, 1.0, 0.1
        out['params'] = np.array([dt, rho, nu, dx, dy])
        for nm, bcs in zip('uvp', (u_bc, v_bc, p_bc)):
            pack_bcs(nm + '_bc', bc_tuples(bcs), out)
        u, v = smooth_field(rng, n, n), smooth_field(rng, n, n)
        p0 = smooth_field(rng, n, n, amp=0.3)
        out.update(u=u, v=v, p0=p0)
        s = M.NavierStokesSystem(None, None, None, u_bc, v_bc, p_bc, nt=1, nit=50, nx=n, ny=n,
                                 dt=dt, rho=rho, nu=nu)
        b = s._build_up_b(u.copy(), v.copy())
        out['b'] = b
        for nit in (1, 50):
            s.nit = nit
            p = p0.copy()
            r = s._pressure_poisson(p, 1e-3 * b)
            out['poisson_nit%d' % nit] = r
        s.nit = 20
        uu, vv, pp = 0.1 * u, 0.1 * v, 0.01 * p0
        a, bb, c = s.step(uu, vv, pp)
        out['step_u'], out['step_v'], out['step_p'] = a, bb, c
        np.savez_compressed(os.path.join(GOLD, 'direct_fd_ops_%d.npz' % n), **out)
    for n, nt in ((16, 10), (64, 10)):
        dx = dy = 2. / (n - 1)
        u_bc, v_bc, p_bc = cavity_bcs(B, dx, dy)
        z = np.zeros((n, n))
        s = M.NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=nt, nit=50, nx=n, ny=n,
                                 dt=1e-3, rho=1, nu=0.1)
        ul, vl, pl = s.simulate()
        out = dict(params=np.array([1e-3, 1.0, 0.1, dx, dy]), nt=np.int64(nt), nit=np.int64(50))
        if n == 16:
            out.update(u=ul, v=vl, p=pl)
        else:
            out.update(u=ul[[0, -1]], v=vl[[0, -1]], p=pl[[0, -1]])
        np.savez_compressed(os.path.join(GOLD, 'direct_fd_cavity_%d.npz' % n), **out)


def worker_chorin_spectral():
    import src.chorin_spectral.simulate as M      # imported FIRST in this process (accommodation 3)
    import src.boundary as B
    rng = np.random.default_rng(0)
    for N in (9, 17, 33, 51):
        d = 2. / (N - 1)
        D = B.DirichletBoundaryCondition
        u_bc = [D(0, 'left', d, d), D(1, 'right', d, d), D(0, 'top', d, d), D(0, 'bottom', d, d)]
        v_bc = [D(0, 'left', d, d), D(0, 'right', d, d), D(0, 'top', d, d), D(0, 'bottom', d, d)]
        dt, rho = 1e-3, 1.0
        s = M.NavierStokesSystem(None, None, None, u_bc, v_bc, nt=1, nit=1, nx=N, ny=N, dt=dt, rho=rho, nu=0.1)
        out = dict(x_i=s.x_i, Dx=s.Dx, Dx_sqr=s.Dx_sqr, DPx=s.DPx, DxDPx=s.DxDPx, Tx=s.Tx, Tx_inv=s.Tx_inv,
                   params=np.array([dt, rho]))
        pack_bcs('u_bc', bc_tuples(u_bc), out)
        pack_bcs('v_bc', bc_tuples(v_bc), out)
        if N in (17, 51):
            un, vn, un1, vn1 = [smooth_field(rng, N, N) for _ in range(4)]
            p = smooth_field(rng, N, N, amp=0.3)
            ui, vi = s._predictor_step(un, vn, un1, vn1)
            a, b, c = s._correction_step(ui, vi, p)
            out.update(un=un, vn=vn, un1=un1, vn1=vn1, p=p, pred_ui=ui, pred_vi=vi,
                       corr_u=a, corr_v=b, corr_p=c)
        np.savez_compressed(os.path.join(GOLD, 'chorin_spectral_%d.npz' % N), **out)


def worker_neural():
    import types
    stub = types.ModuleType('torchdiffeq')           # accommodation 1
    stub.odeint_adjoint = None
    sys.modules['torchdiffeq'] = stub
    import torch
    torch.manual_seed(0)
    import src.neural_spectral.spectral_ode as S1
    import src.neural_spectral.spectral_ode2 as S2
    from src.neural_spectral.anode import odesolver, odesolver_adjoint

    K, nx, ny, nt = 4, 16, 16, 8

    def dump(model):
        return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}

    out = {}
    # integrators at Nt = 7 on an ODEFunc(12)
    f = S1.ODEFunc(12)
    z0 = torch.randn(3, 12)
    for k, v in dump(f).items():
        out['ode_' + k] = v
    out['ode_z0'] = z0.numpy().copy()
    for m in ('Euler', 'RK2', 'RK4'):
        out['ode_' + m] = odesolver(f, z0, {'Nt': 7, 'method': m}).detach().numpy()
    adj = odesolver_adjoint(f, z0, {'Nt': 7, 'method': 'RK4'})
    assert torch.equal(adj, odesolver(f, z0, {'Nt': 7, 'method': 'RK4'}))

    for tag, mod in (('s1', S1), ('s2', S2)):
        model = mod.PDEFunc(K, nx, ny)
        for k, v in dump(model).items():
            out['%s_param_%s' % (tag, k)] = v
        for mb in (1, 3):
            obs = torch.randn(nt, mb, 3, nx, ny)
            t = torch.arange(nt) + 1
            model.zero_grad()
            pred = model(obs[0], t)
            loss = torch.norm(pred - obs, p=2)
            loss.backward()
            pre = '%s_mb%d_' % (tag, mb)
            out[pre + 'obs'] = obs.numpy().copy()
            out[pre + 'pred'] = pred.detach().numpy().copy()
            out[pre + 'loss'] = np.float64(loss.item())
            for n_, p_ in model.named_parameters():
                out[pre + 'grad_' + n_] = p_.grad.detach().numpy().copy()
        if tag == 's1':
            out['s1_diversity_penalty'] = np.float64(model.diversity_penalty().item())
    # BasisFunc per-pixel MLP, forward + grads at 8x8
    bf = S1.BasisFunc(8, 8)
    for k, v in dump(bf).items():
        out['bf_param_' + k] = v
    g = torch.randn(2, 3, 8, 8, requires_grad=True)
    y = bf(g)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    out['bf_in'], out['bf_out'], out['bf_w'] = g.detach().numpy().copy(), y.detach().numpy().copy(), w.numpy().copy()
    out['bf_grad_in'] = g.grad.numpy().copy()
    for n_, p_ in bf.named_parameters():
        out['bf_grad_' + n_] = p_.grad.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, 'neural_spectral.npz'), **out)


def worker_rnn():
    """GRU baselines: src/neural_spectral/spectral_rnn.py PDEFunc (GRU coefficient
    dynamics + basis expansion) and src/neural_spectral/rnn.py RNN (black-box next-frame GRU + MLP)."""
    import types
    stub = types.ModuleType('torchdiffeq')           # accommodation 1 (imported, unused: spectral_rnn.py:10)
    stub.odeint_adjoint = None
    sys.modules['torchdiffeq'] = stub
    import torch
    torch.manual_seed(1)
    import src.neural_spectral.spectral_rnn as SR
    import src.neural_spectral.rnn as R

    def dump(model):
        return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}

    out = {}
    K, nx, ny, nt = 4, 8, 8, 6
    model = SR.PDEFunc(K, nx, ny)
    for k, v in dump(model).items():
        out['sr_param_' + k] = v
    for mb in (1, 2):
        obs = torch.randn(nt, mb, 3, nx, ny)
        t = torch.arange(nt) + 1
        model.zero_grad()
        pred = model(obs[0], t)
        loss = torch.norm(pred - obs, p=2)
        loss.backward()
        pre = 'sr_mb%d_' % mb
        out[pre + 'obs'] = obs.numpy().copy()
        out[pre + 'pred'] = pred.detach().numpy().copy()
        out[pre + 'loss'] = np.float64(loss.item())
        for n_, p_ in model.named_parameters():
            out[pre + 'grad_' + n_] = p_.grad.detach().numpy().copy()
    out['sr_diversity_penalty'] = np.float64(model.diversity_penalty().item())
    rnn = R.RNN(3 * nx * ny, hidden_dim=32)
    for k, v in dump(rnn).items():
        out['rnn_param_' + k] = v
    seq = torch.randn(1, nt, 3 * nx * ny)           # batch 1, as the reference driver (its .view fails on larger batches)
    o, hdn = rnn(seq)
    out['rnn_in'], out['rnn_out'], out['rnn_hid'] = seq.numpy().copy(), o.detach().numpy().copy(), hdn.detach().numpy().copy()
    out['rnn_extrapolate'] = rnn.extrapolate(seq[:1, :1], 4).numpy().copy()
    np.savez_compressed(os.path.join(GOLD, 'neural_rnn.npz'), **out)


def worker_utils():
    """spatial_coarsen: src/utils.py:13-60.  Cases: the default 4x4 blocks, 3x3 = 9
    (pairwise-sum leaf + tail), 2x2 = 4 (short running sum), 16x16 = 256 (one split of the pairwise recursion), and
    agg_x > agg_y (8x2, 5x3, 3x2, 4x2), where the reference's column loop (ny // agg_x) leaves cells at 0; agg_x < agg_y
    overruns the coarse array (recorded as the error the reference raises)."""
    from src.utils import spatial_coarsen
    rng = np.random.default_rng(0)
    out = {}
    for tag, (T, nx, ny, ax, ay) in dict(a=(3, 16, 24, 4, 4), b=(2, 32, 8, 8, 2), c=(2, 20, 9, 5, 3), d=(2, 9, 8, 3, 2),
                                         e=(1, 32, 48, 16, 16), f=(2, 16, 16, 4, 2), g=(2, 9, 12, 3, 3), h=(2, 6, 10, 2, 2)).items():
        X, Y = np.meshgrid(np.linspace(0, 2, nx), np.linspace(0, 2, ny), indexing='ij')
        f = [np.stack([smooth_field(rng, nx, ny) for _ in range(T)]) for _ in range(3)]
        nX, nY, cu, cv, cp = spatial_coarsen(X, Y, f[0], f[1], f[2], agg_x=ax, agg_y=ay)
        out[tag + '_shape'] = np.array([T, nx, ny, ax, ay])
        out[tag + '_u'], out[tag + '_v'], out[tag + '_p'] = f
        out[tag + '_X'], out[tag + '_Y'], out[tag + '_cu'], out[tag + '_cv'], out[tag + '_cp'] = nX, nY, cu, cv, cp
    # agg_x < agg_y: the column loop overruns the coarse array -> the reference raises
    X, Y = np.meshgrid(np.linspace(0, 2, 8), np.linspace(0, 2, 8), indexing='ij')
    z = np.zeros((1, 8, 8))
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            spatial_coarsen(X, Y, z, z, z, agg_x=2, agg_y=4)
        out['overrun_error'] = np.array('none')
    except Exception as e:                                         # noqa: BLE001 -- recording which error the reference raises
        out['overrun_error'] = np.array(type(e).__name__)
    np.savez_compressed(os.path.join(GOLD, 'utils_coarsen.npz'), **out)


WORKERS = dict(utils=worker_utils, rnn=worker_rnn, boundary=worker_boundary, chorin_fd=worker_chorin_fd, direct_fd=worker_direct_fd,
               chorin_spectral=worker_chorin_spectral, neural=worker_neural)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--worker':
        WORKERS[sys.argv[2]]()
        return
    if not os.path.isdir(REF):
        raise SystemExit("reference not present at %s: capture only runs in the build container" % REF)
    os.makedirs(GOLD, exist_ok=True)
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE='1')
    only = [a for a in sys.argv[1:] if a in WORKERS]            # e.g. `python oracle/capture.py rnn` regenerates one fixture
    for name in (only or WORKERS):
        print('capturing', name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', name], check=True, env=env,
                       cwd='/tmp')
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print('golden fixtures: %d files, %.1f KiB' % (len(os.listdir(GOLD)), tot / 1024.))


if __name__ == '__main__':
    main()
