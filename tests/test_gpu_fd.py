"""GPU parity (through the C ABI): boundary conditions, chorin_fd and direct_fd operators vs the
golden vectors captured from the reference and vs the CPU oracle.

Tolerances: float64 kernels keep the reference's operation order (no FMA contraction): stencils
must agree to 1e-13 rel-L2 (typically bitwise), SOR to 1e-12 with the exact sweep count; the
ADI solves go through LAPACK in the reference: 1e-10.  float32 kernels: 1e-5 rel-L2 (north-star).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, unpack_bcs, rel_l2, KINDS, SIDES

pytestmark = pytest.mark.gpu

F64, F32 = 1e-13, 1e-5


def dev(a, dtype=np.float64):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype), device='cuda')


def host(t):
    return t.cpu().numpy()


def objs(bcs):
    from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
    return [(D if k == 'dirichlet' else N)(v, s, dx, dy) for (k, s, v, dx, dy) in bcs]


# ------------------------------------------------------------------ a1 boundary
def test_bc_single_numpy_and_tensor(gpu_device):
    from nns import boundary as B
    g = load_golden('boundary.npz')
    val, dx, dy = float(g['single_value']), float(g['single_dx']), float(g['single_dy'])
    for kind, cls in (('dirichlet', B.DirichletBoundaryCondition), ('neumann', B.NeumannBoundaryCondition)):
        for side in SIDES:
            A = g['A0'].copy()
            r = cls(val, side, dx, dy).apply(A)
            assert r is A                                           # mutates and returns the same object
            np.testing.assert_array_equal(A, g['%s_%s' % (kind, side)])
            T = dev(g['A0'], np.float32)
            r = cls(val, side, dx, dy).apply(T)
            assert r is T
            assert rel_l2(host(T), g['%s_%s' % (kind, side)]) < 1e-6


@pytest.mark.parametrize('name', ['u', 'v', 'p', 'mixed'])
def test_bc_list_corner_order_batched(name, gpu_device):
    from nns import ops
    g = load_golden('boundary.npz')
    bcs = unpack_bcs(g, 'list_%s_bc' % name)
    A = dev(np.stack([g['S0'], 2 * g['S0'], -g['S0']]))
    ops.bc_apply_(A, bcs)
    np.testing.assert_array_equal(host(A[0]), g['list_' + name])
    from oracle.boundary import apply_bc_list
    np.testing.assert_array_equal(host(A[1]), apply_bc_list(2 * g['S0'].copy(), bcs))
    np.testing.assert_array_equal(host(A[2]), apply_bc_list(-g['S0'].copy(), bcs))


def test_bc_non_square_and_errors(gpu_device):
    from nns import ops, _lib
    from oracle.boundary import apply_bc_list
    rng = np.random.default_rng(3)
    A0 = rng.standard_normal((2, 37, 300))
    bcs = [('neumann', 'top', 0.3, 0.1, 0.2), ('dirichlet', 'left', 1.0, 0.1, 0.2), ('neumann', 'right', -2.0, 0.1, 0.2),
           ('neumann', 'bottom', 0.7, 0.1, 0.2)]
    A = dev(A0)
    ops.bc_apply_(A, bcs)
    np.testing.assert_array_equal(host(A), apply_bc_list(A0.copy(), bcs))
    with pytest.raises(ValueError):
        ops.bc_apply_(A, bcs * 3)                                   # > NNS_MAX_BC entries
    with pytest.raises(TypeError):
        ops.bc_apply_(torch.zeros(4, 4, dtype=torch.float64), bcs)  # CPU tensor: no fallback
    with pytest.raises(_lib.NnsError):
        ops.bc_apply_(dev(np.zeros((2, 2))), bcs)                   # nx, ny >= 3


# ------------------------------------------------------------------ a2, a3, a5 chorin_fd stencils
@pytest.mark.parametrize('n', [16, 64])
def test_chorin_fd_predictors_and_correction(n, gpu_device):
    from nns import ops
    g = load_golden('chorin_fd_ops_%d.npz' % n)
    dt, rho, nu, beta, dx, dy = [float(x) for x in g['params']]
    for dtype, tol_s, tol_adi in ((np.float64, F64, 1e-10), (np.float32, F32, F32)):
        u, v, u1, v1, p0 = (dev(g[k], dtype) for k in ('u', 'v', 'u1', 'v1', 'p0'))
        ui, vi = ops.fd_predictor_explicit(u, v, u1, v1, dt, dx, dy, nu)
        assert rel_l2(host(ui), g['pred_explicit_ui']) <= tol_s and rel_l2(host(vi), g['pred_explicit_vi']) <= tol_s
        ui, vi = ops.fd_predictor_adi(u, v, u1, v1, dt, dx, dy, nu)
        assert rel_l2(host(ui), g['pred_semi_implicit_ui']) <= tol_adi and rel_l2(host(vi), g['pred_semi_implicit_vi']) <= tol_adi
        a, b = ops.fd_correction(u, v, p0, dt, dx, dy)
        assert rel_l2(host(a), g['corr_u']) <= tol_s and rel_l2(host(b), g['corr_v']) <= tol_s


def test_chorin_fd_f64_explicit_is_bitwise(gpu_device):
    from nns import ops
    g = load_golden('chorin_fd_ops_64.npz')
    dt, rho, nu, beta, dx, dy = [float(x) for x in g['params']]
    ui, vi = ops.fd_predictor_explicit(dev(g['u']), dev(g['v']), dev(g['u1']), dev(g['v1']), dt, dx, dy, nu)
    np.testing.assert_array_equal(host(ui), g['pred_explicit_ui'])
    np.testing.assert_array_equal(host(vi), g['pred_explicit_vi'])


def test_adi_rejects_non_square(gpu_device):
    from nns import ops, _lib
    z = dev(np.zeros((8, 12)))
    with pytest.raises(_lib.NnsError, match='nx == ny'):
        ops.fd_predictor_adi(z, z, z, z, 1e-3, 0.1, 0.1, 0.1)


@pytest.mark.parametrize('shape', [(3, 33, 70), (2, 130, 45)])
def test_stencils_vs_oracle_ragged_batched(shape, gpu_device):
    """Non-square, non-multiple-of-anything sizes, batch > 1, against the oracle."""
    from nns import ops
    from oracle import chorin_fd as OC, direct_fd as OD
    rng = np.random.default_rng(11)
    u, v, u1, v1, p = (rng.standard_normal(shape) for _ in range(5))
    dt, dx, dy, nu, rho = 1e-3, 2. / (shape[1] - 1), 2. / (shape[2] - 1), 0.05, 1.3
    ui, vi = ops.fd_predictor_explicit(dev(u), dev(v), dev(u1), dev(v1), dt, dx, dy, nu)
    a, b = OC.explicit_predictor(u, v, u1, v1, dt, dx, dy, nu)
    assert rel_l2(host(ui), a) <= F64 and rel_l2(host(vi), b) <= F64
    C = ops.fd_pressure_rhs(dev(u), dev(v), dt, dx, dy, rho)
    assert rel_l2(host(C), OC.pressure_rhs(u, v, dt, dx, dy, rho)) <= F64
    a, b = ops.fd_correction(dev(u), dev(v), dev(p), dt, dx, dy)
    ra, rb = OC.correction(u, v, p, dt, dx, dy)
    assert rel_l2(host(a), ra) <= F64 and rel_l2(host(b), rb) <= F64
    bb = ops.fd_build_b(dev(u), dev(v), dt, dx, dy, rho)
    assert rel_l2(host(bb), OD.build_up_b(u, v, dt, dx, dy, rho)) <= F64
    a, b = ops.fd_direct_update(dev(u), dev(v), dev(p), dt, dx, dy, rho, nu)
    ra, rb = OD.momentum_update(u, v, p, dt, dx, dy, rho, nu)
    assert rel_l2(host(a), ra) <= F64 and rel_l2(host(b), rb) <= F64


# ------------------------------------------------------------------ a4 SOR
@pytest.mark.parametrize('n', [16, 64])
@pytest.mark.parametrize('nit', [3, 50, 400])
def test_sor_matches_reference_and_sweep_count(n, nit, gpu_device):
    from nns import ops
    from nns.chorin_fd import SOR_TOL
    g = load_golden('chorin_fd_ops_%d.npz' % n)
    dt, rho, nu, beta, dx, dy = [float(x) for x in g['params']]
    C = ops.fd_pressure_rhs(dev(g['press_ui']), dev(g['press_vi']), dt, dx, dy, rho)
    p = dev(g['press_p0_nit%d' % nit])
    info = host(ops.fd_sor_(p, C, dx, dy, beta, SOR_TOL, nit - 1))
    assert int(info[0, 0]) == int(g['press_sweeps_nit%d' % nit])
    assert rel_l2(host(p), g['press_p_nit%d' % nit]) <= 1e-12
    assert abs(info[0, 1] - float(g['press_err_nit%d' % nit])) <= 1e-12 * max(1.0, abs(float(g['press_err_nit%d' % nit])))
    # float32: fields within 1e-5, sweep count may differ by one at the tolerance boundary
    C32 = ops.fd_pressure_rhs(dev(g['press_ui'], np.float32), dev(g['press_vi'], np.float32), dt, dx, dy, rho)
    p32 = dev(g['press_p0_nit%d' % nit], np.float32)
    info32 = host(ops.fd_sor_(p32, C32, dx, dy, beta, SOR_TOL, nit - 1))
    assert abs(int(info32[0, 0]) - int(g['press_sweeps_nit%d' % nit])) <= (1 if nit == 400 else 0)
    assert rel_l2(host(p32), g['press_p_nit%d' % nit]) <= 2e-5


def test_sor_batched_replicas_large_global_memory_path_and_determinism(gpu_device):
    """Batch of different grids (one stops early, one does not), a grid too large for the LDS-resident
    path (global-memory variant), and run-to-run bitwise determinism (catches LDS/pipeline races)."""
    from nns import ops
    from oracle import chorin_fd as OC
    rng = np.random.default_rng(2)
    n = 150                                                 # 2 * 150^2 * 8 B > 150 KB -> global path in f64
    dx = dy = 2. / (n - 1)
    P0 = np.stack([1e-7 * rng.standard_normal((n, n)), 0.5 * rng.standard_normal((n, n))])
    Cc = np.stack([np.zeros((n, n)), 1e-2 * rng.standard_normal((n, n))])
    for c in Cc:
        c[0, :] = c[-1, :] = 0; c[:, 0] = c[:, -1] = 0
    outs = []
    for rep in range(2):
        p = dev(P0)
        info = host(ops.fd_sor_(p, dev(Cc), dx, dy, 1.25, 5e-6, 40))
        outs.append((host(p), info))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    for b in range(2):
        pr = P0[b].copy()
        err, sweeps = 1.0, 0
        prev = pr.copy()
        while err > 5e-6 and sweeps < 40:
            OC.sor_sweep_wavefront(pr, Cc[b], dx, dy, 1.25)
            err = np.abs(pr - prev).max(); prev = pr.copy(); sweeps += 1
        assert int(outs[0][1][b, 0]) == sweeps
        assert rel_l2(outs[0][0][b], pr) <= 1e-12
    assert int(outs[0][1][0, 0]) < 40 and int(outs[0][1][1, 0]) == 40


# ------------------------------------------------------------------ a6 step / simulate through the reference call surface
@pytest.mark.parametrize('n', [16, 64])
@pytest.mark.parametrize('method', ['explicit', 'semi_implicit'])
def test_chorin_fd_step_surface(n, method, gpu_device):
    from src.chorin_fd.simulate import NavierStokesSystem
    g = load_golden('chorin_fd_ops_%d.npz' % n)
    dt, rho, nu, beta, dx, dy = [float(x) for x in g['params']]
    u_bc, v_bc, p_bc = (objs(unpack_bcs(g, k + '_bc')) for k in 'uvp')
    s = NavierStokesSystem(None, None, None, u_bc, v_bc, p_bc, nt=1, nit=20, nx=n, ny=n, dt=dt, rho=rho, nu=nu,
                           beta=beta, method=method)
    p = 0.01 * g['p0'].copy()
    a, b, c = s.step(0.1 * g['u'], 0.1 * g['v'], 0.1 * g['u1'], 0.1 * g['v1'], p)
    assert c is p                                              # p mutated in place and returned
    tol = 1e-12 if method == 'explicit' else 1e-9
    for got, key in ((a, 'u'), (b, 'v'), (c, 'p')):
        assert rel_l2(got, g['step_%s_%s' % (method, key)]) <= tol
    assert s.sor_info()[0][0] == int(g['step_%s_sweeps' % method])
    with pytest.raises(AssertionError):
        NavierStokesSystem(None, None, None, u_bc, v_bc, p_bc, method='implicit')


@pytest.mark.parametrize('case', [(16, 'explicit', 0.1), (16, 'semi_implicit', 0.1), (64, 'explicit', 0.02),
                                  (64, 'explicit', 0.01), (64, 'semi_implicit', 0.02), (64, 'semi_implicit', 0.01)])
def test_chorin_fd_cavity_trajectory(case, gpu_device):
    """BASELINE config 1 (64x64 lid-driven cavity, Re = 100) and the 16x16 case, f64 and f32."""
    from src.chorin_fd.simulate import NavierStokesSystem
    from src.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
    n, method, nu = case
    g = load_golden('chorin_fd_cavity_%d_%s_nu%g.npz' % (n, method, nu))
    dt, rho, nu_, beta, dx, dy = [float(x) for x in g['params']]
    u_bc = [D(0, 'left', dx, dy), D(1, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    v_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
    p_bc = [D(0, 'top', dx, dy), N(0, 'bottom', dx, dy), N(0, 'left', dx, dy), N(0, 'right', dx, dy)]
    sel = slice(None) if n == 16 else [0, -1]
    for dtype, tol in ((np.float64, 1e-9), (np.float32, 2e-5)):
        z = np.zeros((n, n))
        s = NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=int(g['nt']), nit=int(g['nit']), nx=n, ny=n,
                               dt=dt, rho=rho, nu=nu_, beta=beta, method=method, dtype=dtype)
        ul, vl, pl = s.simulate()
        assert ul.dtype == np.float64 and ul.shape == (int(g['nt']), n, n)
        assert rel_l2(ul[sel], g['u']) <= tol and rel_l2(vl[sel], g['v']) <= tol and rel_l2(pl[sel], g['p']) <= tol


def test_chorin_fd_ensemble_batch_equals_singles(gpu_device):
    from nns.chorin_fd import NavierStokesSystem
    from oracle.boundary import cavity_bcs
    n = 32
    dx = dy = 2. / (n - 1)
    u_bc, v_bc, p_bc = cavity_bcs(dx, dy)
    rng = np.random.default_rng(9)
    U = 0.05 * rng.standard_normal((3, n, n))
    kw = dict(nt=3, nit=30, nx=n, ny=n, dt=1e-3, rho=1, nu=0.05, beta=1.25, method='explicit')
    ub, vb, pb = NavierStokesSystem(U, 0 * U, 0 * U, u_bc, v_bc, p_bc, **kw).simulate()
    assert ub.shape == (3, 3, n, n)
    for b in range(3):
        u1, v1, p1 = NavierStokesSystem(U[b], 0 * U[b], 0 * U[b], u_bc, v_bc, p_bc, **kw).simulate()
        np.testing.assert_array_equal(ub[:, b], u1)
        np.testing.assert_array_equal(pb[:, b], p1)


# ------------------------------------------------------------------ a7-a9 direct_fd
@pytest.mark.parametrize('n', [16, 64])
def test_direct_fd_operators_and_step(n, gpu_device):
    from src.direct_fd.simulate import NavierStokesSystem
    g = load_golden('direct_fd_ops_%d.npz' % n)
    dt, rho, nu, dx, dy = [float(x) for x in g['params']]
    u_bc, v_bc, p_bc = (objs(unpack_bcs(g, k + '_bc')) for k in 'uvp')
    s = NavierStokesSystem(None, None, None, u_bc, v_bc, p_bc, nt=1, nit=50, nx=n, ny=n, dt=dt, rho=rho, nu=nu)
    b = s._build_up_b(g['u'], g['v'])
    assert rel_l2(b, g['b']) <= F64
    for nit in (1, 50):
        s.nit = nit
        p = g['p0'].copy()
        r = s._pressure_poisson(p, 1e-3 * g['b'])
        assert r is p
        assert rel_l2(p, g['poisson_nit%d' % nit]) <= F64
    s.nit = 20
    u, v, p = 0.1 * g['u'], 0.1 * g['v'], 0.01 * g['p0']
    ru, rv, rp = s.step(u, v, p)
    assert ru is u and rv is v and rp is p                           # in place, as the reference
    assert rel_l2(u, g['step_u']) <= F64 and rel_l2(v, g['step_v']) <= F64 and rel_l2(p, g['step_p']) <= F64


def test_direct_fd_jacobi_large_grid_multilaunch_path(gpu_device):
    """A grid too large for the LDS-resident Jacobi: ping-pong launches + BC kernel, odd and even nit."""
    from nns import ops
    from oracle import direct_fd as OD
    rng = np.random.default_rng(4)
    n, m = 140, 170
    dx, dy = 2. / (n - 1), 2. / (m - 1)
    p0, b = rng.standard_normal((2, n, m)), rng.standard_normal((2, n, m))
    bcs = [('dirichlet', 'top', 0.0, dx, dy), ('neumann', 'bottom', 0.2, dx, dy), ('neumann', 'left', 0.0, dx, dy),
           ('neumann', 'right', -0.1, dx, dy)]
    for nit in (3, 4):
        p = dev(p0)
        ops.fd_jacobi_(p, dev(b), dx, dy, nit, bcs)
        ref = p0.copy()
        for k in range(2):
            OD.pressure_poisson(ref[k], b[k], bcs, dx, dy, nit)
        assert rel_l2(host(p), ref) <= F64


@pytest.mark.parametrize('n', [16, 64])
def test_direct_fd_cavity_trajectory(n, gpu_device):
    from src.direct_fd.simulate import NavierStokesSystem
    from oracle.boundary import cavity_bcs
    g = load_golden('direct_fd_cavity_%d.npz' % n)
    dt, rho, nu, dx, dy = [float(x) for x in g['params']]
    u_bc, v_bc, p_bc = (objs(b) for b in cavity_bcs(dx, dy))
    sel = slice(None) if n == 16 else [0, -1]
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
        z = np.zeros((n, n))
        u_ic = z.copy()
        s = NavierStokesSystem(u_ic, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=int(g['nt']), nit=int(g['nit']), nx=n, ny=n,
                               dt=dt, rho=rho, nu=nu, dtype=dtype)
        ul, vl, pl = s.simulate()
        assert rel_l2(ul[sel], g['u']) <= tol and rel_l2(vl[sel], g['v']) <= tol and rel_l2(pl[sel], g['p']) <= tol
        assert np.abs(u_ic).max() > 0                        # the reference's simulate mutates the caller's ICs (:132)


def test_graph_replay_equals_eager_loop(gpu_device):
    """simulate_device(use_graph=True) captures one step in a hipGraph: results must be bitwise those of the eager loop."""
    from nns.chorin_fd import NavierStokesSystem
    from nns.direct_fd import NavierStokesSystem as DirectNS
    from oracle.boundary import cavity_bcs
    n = 32
    dx = dy = 2. / (n - 1)
    u_bc, v_bc, p_bc = cavity_bcs(dx, dy)
    z = np.zeros((n, n))
    for method in ('explicit', 'semi_implicit'):
        s = NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=12, nit=30, nx=n, ny=n, dt=1e-3, rho=1, nu=0.05, beta=1.25, method=method)
        a = s.simulate_device(use_graph=True)
        b = s.simulate_device(use_graph=False)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    d = DirectNS(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=12, nit=20, nx=n, ny=n, dt=1e-3, rho=1, nu=0.1)
    ug = d.simulate_device(use_graph=True)[0]
    d2 = DirectNS(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=12, nit=20, nx=n, ny=n, dt=1e-3, rho=1, nu=0.1)
    assert torch.equal(ug, d2.simulate_device(use_graph=False)[0])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_corrected_options_vs_oracle(gpu_device, dtype):
    """SURVEY section 8 (f) rank 3 options: the corrected explicit predictor (1e-13 / 1e-5 vs the oracle) and the
    red-black SOR (bitwise in float64 incl. sweep count; early stop and sweep cap; ragged and batched grids)."""
    import torch
    from nns import ops
    from oracle import chorin_fd as O
    rng = np.random.default_rng(12)
    tol = 1e-13 if dtype == "float64" else 2e-5
    for shape in ((1, 33, 47), (3, 64, 64)):
        f = [rng.standard_normal(shape).astype(dtype) for _ in range(4)]
        dt, dx, dy, nu = 1e-3, 0.03, 0.05, 0.1
        ref = O.explicit_predictor_corrected(*[a.astype(np.float64) for a in f], dt, dx, dy, nu)
        got = ops.fd_predictor_explicit_corrected(*[torch.as_tensor(a, device="cuda") for a in f], dt, dx, dy, nu)
        for g, r in zip(got, ref):
            assert rel_l2(g.cpu().numpy(), r) < tol
    # the last three do not fit LDS: chained chip-wide half-sweeps, switched off on the device (sweep cap; early stop at
    # sweep ~22 of 60; a tolerance above the initial err = 1 -> no sweep at all)
    for (nx, ny), cap, stol in (((20, 17), 30, 5e-6), ((64, 64), 49, 5e-6), ((40, 56), 2000, 1e-4), ((150, 140), 12, 5e-6),
                                ((160, 170), 60, 0.05), ((300, 129), 3, 2.0)):
        B = 2
        C = (rng.standard_normal((B, nx, ny)) * (1e-5 if stol == 0.05 else 0.1)).astype(dtype)
        p0 = (rng.standard_normal((B, nx, ny)) * 0.01).astype(dtype)
        dx, dy, beta = 1.0 / nx, 1.0 / ny, 1.5
        p = torch.as_tensor(p0.copy(), device="cuda")
        info = ops.fd_sor_redblack_(p, torch.as_tensor(C, device="cuda"), dx, dy, beta, stol, cap).cpu().numpy()
        for b in range(B):
            pr = p0[b].copy(); err, sweeps = 1, 0
            prev = pr.copy()
            while err > stol and sweeps < cap:
                O.sor_sweep_redblack(pr, C[b], np.dtype(dtype).type(dx), np.dtype(dtype).type(dy), np.dtype(dtype).type(beta))
                err = np.max(np.abs(pr - prev)); prev = pr.copy(); sweeps += 1
            if stol == 0.05:
                assert 5 < sweeps < cap                                # the early stop is what this case exercises
            if stol == 2.0:
                assert sweeps == 0
            if dtype == "float64":
                assert int(info[b, 0]) == sweeps, (nx, ny, b)
                assert np.array_equal(p[b].cpu().numpy(), pr), (nx, ny, b)
                assert info[b, 1] == err
            else:
                assert abs(int(info[b, 0]) - sweeps) <= 1
                assert rel_l2(p[b].cpu().numpy(), pr) < 1e-4




@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_sharded_redblack_halfsweep(gpu_device, dtype):
    """The multi-workgroup half-sweep kernel behind nns.slab.SlabPressure: (1) one rank through SlabPressure == the
    oracle's red-black solve (bitwise + same sweep count in float64); (2) a hand-made 3-slab split in one process (odd
    row offsets, so the colour parity depends on gi0) with halo copies == the same."""
    import torch
    import torch.distributed as dist
    from nns import ops
    from nns.slab import SlabPressure
    from oracle import chorin_fd as O
    rng = np.random.default_rng(77)
    T = np.dtype(dtype).type
    nx, ny, beta, cap, stol = 301, 517, 1.5, 25, 5e-6
    dx, dy = 1.0 / nx, 1.0 / ny
    C = (rng.standard_normal((nx, ny)) * 0.1).astype(dtype)
    p0 = (rng.standard_normal((nx, ny)) * 0.01).astype(dtype)
    ref = p0.copy(); prev = ref.copy(); err, sweeps = 1, 0
    while err > stol and sweeps < cap:
        O.sor_sweep_redblack(ref, C, T(dx), T(dy), T(beta))
        err = np.max(np.abs(ref - prev)); prev = ref.copy(); sweeps += 1

    def check(got, done=None):
        if dtype == "float64":
            assert np.array_equal(got, ref)
            assert done is None or done == sweeps
        else:
            assert rel_l2(got, ref) < 1e-4

    created = False
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29919', rank=0, world_size=1)
        created = True
    try:
        s = SlabPressure(nx, ny, dx, dy, beta, tol=stol)
        p = torch.as_tensor(p0.copy(), device="cuda")
        done, e = s.solve_(p, torch.as_tensor(C, device="cuda"), cap)
        check(p.cpu().numpy(), done)
        if dtype == "float64":
            assert e == err
    finally:
        if created:
            dist.destroy_process_group()

    cuts = [0, 101, 198, nx]                                               # slabs of 101, 97, 103 rows: gi0 = 0, 100, 197
    slabs, Cs, gi0s = [], [], []
    for r in range(3):
        lo, hi = cuts[r] - (r > 0), cuts[r + 1] + (r < 2)
        slabs.append(torch.as_tensor(p0[lo:hi].copy(), device="cuda")); Cs.append(torch.as_tensor(C[lo:hi].copy(), device="cuda")); gi0s.append(lo)
    ebuf = torch.zeros(1, dtype=slabs[0].dtype, device="cuda")
    for _ in range(sweeps):
        for colour in (0, 1):
            for r in range(2):                                             # halo refresh: what SlabPressure._exchange does over the ring
                slabs[r][-1].copy_(slabs[r + 1][1]); slabs[r + 1][0].copy_(slabs[r][-2])
            for r in range(3):
                ops.fd_sor_redblack_halfsweep_(slabs[r], Cs[r], ebuf, gi0s[r], colour, dx, dy, beta)
    got = torch.cat([slabs[0][:-1], slabs[1][1:-1], slabs[2][1:]]).cpu().numpy()
    check(got)


def test_slab_chorin_fd_single_rank_uses_hip_backend(gpu_device):
    """nns.slab.SlabChorinFD with its default (HIP) operators on one rank == the single-GPU driver with
    pressure_solver='redblack' (bitwise, float64) and the oracle (1e-9); halo logic: tests/test_slab_gloo.py."""
    import torch
    import torch.distributed as dist
    from nns.chorin_fd import NavierStokesSystem
    from nns.slab import SlabChorinFD
    from oracle import chorin_fd as O
    from oracle.boundary import cavity_bcs
    n, nt = 48, 6
    dx = dy = 2. / (n - 1)
    u_bc, v_bc, p_bc = cavity_bcs(dx, dy)
    z = np.zeros((n, n))
    created = False
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29921', rank=0, world_size=1)
        created = True
    try:
        for adv in ('reference', 'corrected'):
            s = SlabChorinFD(u_bc, v_bc, p_bc, 50, n, n, 1e-3, 1.0, 0.05, 1.25, advection=adv)
            us, vs, ps = s.simulate(*[torch.as_tensor(z.copy(), device='cuda') for _ in range(3)], nt)
            u, v, p = NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=nt, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.05,
                                         beta=1.25, method='explicit', advection=adv, pressure_solver='redblack').simulate()
            for g, r in zip((us, vs, ps), (u, v, p)):
                assert np.array_equal(g.cpu().numpy(), r), adv
            ur, vr, pr = O.simulate(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt, 50, 1e-3, 1, 0.05, 1.25, 'explicit',
                                    advection=adv, pressure_solver='redblack')
            assert np.abs(us.cpu().numpy() - ur).max() < 1e-9 and np.abs(ps.cpu().numpy() - pr).max() < 1e-9
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_adi_predictor_on_column_slab(gpu_device, dtype):
    """nns_fd_predictor_adi_colslab_*: the reference ADI on a column slab [nx, nyl] of a square grid equals the same
    columns of the full-grid result (the solves run along axis 0) and the oracle on the slab; and SlabChorinFD
    (method='semi_implicit') on one rank equals the single-GPU driver with the red-black pressure option."""
    import torch
    import torch.distributed as dist
    from nns import ops
    from nns.chorin_fd import NavierStokesSystem
    from nns.slab import SlabChorinFD
    from oracle import chorin_fd as O
    from oracle.boundary import cavity_bcs
    rng = np.random.default_rng(8)
    n = 40
    f = [rng.standard_normal((n, n)).astype(dtype) for _ in range(4)]
    dt, h, nu = 1e-2, 2. / (n - 1), 0.2
    tol = 1e-12 if dtype == "float64" else 3e-5
    full = ops.fd_predictor_adi(*[torch.as_tensor(a, device="cuda") for a in f], dt, h, h, nu)
    lo, hi = 9, 26                                                        # slab = columns 9..25: owned 10..24 + one halo column each side
    sl = [np.ascontiguousarray(a[:, lo:hi]) for a in f]
    got = ops.fd_predictor_adi(*[torch.as_tensor(a, device="cuda") for a in sl], dt, h, h, nu, column_slab=True)
    ref = O.semi_implicit_predictor(*[a.astype(np.float64) for a in sl], dt, h, h, nu, column_slab=True)
    for g, r, fu in zip(got, ref, full):
        assert rel_l2(g.cpu().numpy(), r) < tol
        assert torch.equal(g[:, 1:-1], fu[:, lo + 1:hi - 1])             # interior columns: bitwise the full-grid run
    with pytest.raises(RuntimeError):
        ops.fd_predictor_adi(*[torch.as_tensor(a, device="cuda") for a in sl], dt, h, h, nu)      # not square, not declared a slab
    if dtype == "float64":
        created = False
        if not dist.is_initialized():
            dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29923', rank=0, world_size=1)
            created = True
        try:
            u_bc, v_bc, p_bc = cavity_bcs(h, h)
            z = np.zeros((n, n))
            s = SlabChorinFD(u_bc, v_bc, p_bc, 50, n, n, 1e-3, 1.0, 0.05, 1.25, method='semi_implicit')
            us, vs, ps = s.simulate(*[torch.as_tensor(z.copy(), device='cuda') for _ in range(3)], 5)
            u, v, p = NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=5, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.05,
                                         beta=1.25, method='semi_implicit', pressure_solver='redblack').simulate()
            for g, r in zip((us, vs, ps), (u, v, p)):
                assert np.array_equal(g.cpu().numpy(), r)
        finally:
            if created:
                dist.destroy_process_group()


def test_cavity_with_corrected_options(gpu_device):
    """The cavity driver with advection='corrected', pressure_solver='redblack' against the oracle run with the same
    options (float64, 1e-9); the defaults still reproduce the reference."""
    from nns.chorin_fd import NavierStokesSystem
    from oracle import chorin_fd as O
    from oracle.boundary import cavity_bcs
    n, nt = 32, 8
    dx = dy = 2. / (n - 1)
    u_bc, v_bc, p_bc = cavity_bcs(dx, dy)
    z = np.zeros((n, n))
    kw = dict(nt=nt, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.05, beta=1.25, method='explicit')
    u, v, p = NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, advection='corrected', pressure_solver='redblack', **kw).simulate()
    ur, vr, pr = O.simulate(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt, 50, 1e-3, 1, 0.05, 1.25, 'explicit',
                            advection='corrected', pressure_solver='redblack')
    assert np.abs(u - ur).max() < 1e-9 and np.abs(v - vr).max() < 1e-9 and np.abs(p - pr).max() < 1e-9
    u0, _, _ = NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, **kw).simulate()
    assert np.abs(u0 - u).max() > 1e-6                                  # the options do change the flow


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_corrected_adi_vs_oracle(gpu_device, dtype):
    """True y-direction ADI (LDS-tiled row-wise Thomas solve) vs the oracle: square, non-square, ragged (not multiples of
    the 64-wide tiles) and batched grids; and through the driver (method='semi_implicit', advection='corrected')."""
    from nns import ops
    from nns.chorin_fd import NavierStokesSystem
    from oracle import chorin_fd as O
    from oracle.boundary import cavity_bcs
    rng = np.random.default_rng(31)
    tol = 1e-12 if dtype == "float64" else 3e-5
    for shape in ((1, 24, 24), (2, 33, 70), (1, 130, 65), (3, 64, 128)):
        f = [rng.standard_normal(shape).astype(dtype) for _ in range(4)]
        dt, dx, dy, nu = 1e-2, 0.05, 0.03, 0.2
        ref = O.semi_implicit_predictor_corrected(*[a.astype(np.float64) for a in f], dt, dx, dy, nu)
        got = ops.fd_predictor_adi(*[torch.as_tensor(a, device="cuda") for a in f], dt, dx, dy, nu, corrected=True)
        for g, r in zip(got, ref):
            assert rel_l2(g.cpu().numpy(), r) < tol, shape
    if dtype == "float64":
        n, nt = 32, 6
        dx = dy = 2. / (n - 1)
        u_bc, v_bc, p_bc = cavity_bcs(dx, dy)
        z = np.zeros((n, n))
        kw = dict(nt=nt, nit=50, nx=n, ny=n, dt=1e-3, rho=1, nu=0.05, beta=1.25, method='semi_implicit')
        u, v, p = NavierStokesSystem(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, advection='corrected', **kw).simulate()
        ur, vr, pr = O.simulate(z.copy(), z.copy(), z.copy(), u_bc, v_bc, p_bc, nt, 50, 1e-3, 1, 0.05, 1.25, 'semi_implicit', advection='corrected')
        assert np.abs(u - ur).max() < 1e-9 and np.abs(p - pr).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('n,B,dtype,advection,nit', [(64, 1, np.float64, 'reference', 50), (51, 1, np.float64, 'reference', 50), (33, 3, np.float32, 'corrected', 20),
                                                      (64, 2, np.float64, 'corrected', 9), (17, 1, np.float64, 'reference', 200), (96, 2, np.float32, 'reference', 30), (80, 1, np.float64, 'reference', 15)])
def test_fused_explicit_step_is_bitwise_the_separate_operators(n, B, dtype, advection, nit, gpu_device):
    """Round 4: chorin_fd's explicit step as ONE launch (nns_fd_step_explicit_*: predictor, boundary lists, right-hand side, lexicographic SOR,
    pressure boundary list, correction by one workgroup per grid) against the seven separate launches -- the same per-point functions, so u, v, p,
    the sweep counts and the last errors must agree BITWISE over a cavity run (early stops of the solve included once the flow has developed),
    single grids and batches, both advection forms, float64 and float32; the C entry refuses aliased outputs and grids that do not fit LDS."""
    from nns import ops, _lib
    from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
    from nns.chorin_fd import NavierStokesSystem
    dx = dy = 2. / (n - 1)
    u_bc = [D(0., 'left', dx, dy), D(1., 'right', dx, dy), D(0., 'top', dx, dy), D(0., 'bottom', dx, dy)]
    v_bc = [D(0., 'left', dx, dy), D(0., 'right', dx, dy), N(0.5, 'top', dx, dy), D(0., 'bottom', dx, dy)]
    p_bc = [D(0., 'top', dx, dy), N(0., 'bottom', dx, dy), N(0.25, 'left', dx, dy), N(0., 'right', dx, dy)]
    rng = np.random.default_rng(n + B)
    shape = (n, n) if B == 1 else (B, n, n)
    ic = [0.05 * rng.standard_normal(shape) for _ in range(3)]
    runs = []
    for fused in (True, False):
        s = NavierStokesSystem(ic[0].copy(), ic[1].copy(), ic[2].copy(), u_bc, v_bc, p_bc, nt=12, nit=nit, nx=n, ny=n, dt=1e-3, rho=1.1, nu=0.02, beta=1.25,
                               method='explicit', dtype=dtype, advection=advection)
        s.fused_step = fused
        assert s._fused_step_applies(s._d(ic[2])) == fused
        us, vs, ps = s.simulate_device()
        runs.append((us, vs, ps, s.last_sor_info.clone()))
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    assert float(runs[0][0].abs().max()) > 0.5                     # the lid moves the fluid
    # the C entry's refusals
    f = [torch.zeros(1, n, n, dtype=torch.float64 if dtype == np.float64 else torch.float32, device='cuda') for _ in range(6)]
    bl = ops.make_bc_list(u_bc)
    with pytest.raises(_lib.NnsError, match='must not be input'):
        ops.fd_step_explicit(f[0], f[1], f[2], f[3], f[4], bl, bl, bl, 1e-3, dx, dy, 1.0, 0.02, 1.25, 1e-3, 5, out=(f[0], f[5]))
    big = [torch.zeros(1, 200, 200, dtype=torch.float64, device='cuda') for _ in range(5)]
    assert not ops.fd_step_explicit_fits(200, 200, torch.float64)
    with pytest.raises(_lib.NnsError, match='does not fit'):
        ops.fd_step_explicit(*big, bl, bl, bl, 1e-3, dx, dy, 1.0, 0.02, 1.25, 1e-3, 5)


@pytest.mark.gpu
@pytest.mark.parametrize('n,dtype', [(64, torch.float64), (51, torch.float32), (80, torch.float64)])
def test_sor_result_does_not_depend_on_the_sweep_count_hint(n, dtype, gpu_device):
    """nns_fd_sor_hint_*: the previous solve's sweep count only sizes the first speculative batch of sweeps.  Whatever it says (right, too small,
    too large, absurd), p, the sweep count and the last error are BITWISE those of the un-hinted solve -- early stops inside a batch included."""
    from nns import ops
    rng = np.random.default_rng(n)
    for nit, tol in ((49, 1e-3), (49, 0.3), (120, 0.05), (3, 1e-9)):
        p0 = torch.as_tensor(rng.standard_normal((2, n, n)), dtype=dtype, device='cuda')
        C = torch.as_tensor(rng.standard_normal((2, n, n)) * 5, dtype=dtype, device='cuda')
        ref = p0.clone()
        info_ref = ops.fd_sor_(ref, C, 0.03, 0.04, 1.25, tol, nit)
        for h in (1, 2, int(info_ref[0, 0].item()), int(info_ref[0, 0].item()) + 1, 49, 1000, 0, -7):
            hint = torch.tensor([[h, 0.0], [max(h - 1, 0), 0.0]], dtype=dtype, device='cuda')
            p = p0.clone()
            info = ops.fd_sor_(p, C, 0.03, 0.04, 1.25, tol, nit, hint=hint)
            assert torch.equal(p, ref) and torch.equal(info, info_ref), (nit, tol, h)
        p = p0.clone()                                   # the hint may be the info buffer itself (read before it is written)
        info = info_ref.clone()
        from nns import _lib
        nbytes = _lib.lib().nns_fd_sor_workspace(2, n, n, p.element_size())
        work = torch.empty(nbytes // p.element_size(), dtype=dtype, device='cuda')
        fn = getattr(_lib.lib(), 'nns_fd_sor_hint_f64' if dtype == torch.float64 else 'nns_fd_sor_hint_f32')
        assert fn(p.data_ptr(), C.data_ptr(), info.data_ptr(), info.data_ptr(), work.data_ptr(), 2, n, n, 0.03, 0.04, 1.25, tol, nit, torch.cuda.current_stream().cuda_stream) == 0
        assert torch.equal(p, ref) and torch.equal(info, info_ref)


@pytest.mark.gpu
@pytest.mark.parametrize('nx,ny,B,dtype,corrected', [(64, 64, 1, torch.float64, False), (51, 51, 3, torch.float64, False), (33, 33, 2, torch.float32, False),
                                                      (64, 64, 2, torch.float64, True), (40, 56, 1, torch.float32, True), (96, 96, 1, torch.float64, False)])
def test_adi_predictor_lds_kernel_is_bitwise_the_streaming_kernel(nx, ny, B, dtype, corrected, gpu_device, monkeypatch):
    """Round 4: the semi-implicit predictor of grids that fit one workgroup's LDS (right-hand sides chip-wide, the factorisation tabulated once and
    cut at its floating-point fixed point, recurrences on LDS operands, divisions as Markstein's exact form) against the streaming kernel it
    replaces there: the same expressions in the same order, so ui, vi agree BITWISE -- both advection forms, batches, float32 / float64, fields at
    rest (zeros stay zeros)."""
    import subprocess, sys, os, tempfile
    from conftest import PKG, ROOT
    code = '''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nns import ops
rng = np.random.default_rng(3)
dt_ = torch.float64 if %r == 'float64' else torch.float32
f = [torch.as_tensor(rng.standard_normal((%d, %d, %d)) * s, dtype=dt_, device='cuda') for s in (1.0, 1.0, 0.9, 0.9)]
outs = []
for fields in (f, [torch.zeros_like(t) for t in f]):
    ui, vi = ops.fd_predictor_adi(*fields, 1e-3, 0.03, 0.04 if %r else 0.03, 0.02, corrected=%r)
    outs += [ui.cpu().numpy(), vi.cpu().numpy()]
np.savez(sys.argv[1], *outs)
''' % (PKG, str(dtype)[6:], B, nx, ny, corrected, corrected)
    res = []
    with tempfile.TemporaryDirectory() as d:
        for flag in ('1', '0'):
            out = os.path.join(d, 'o%s.npz' % flag)
            r = subprocess.run([sys.executable, '-c', code, out], env=dict(os.environ, NNS_ADI_LDS=flag), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            z = np.load(out)
            res.append([z[k] for k in z.files])
    for a, b in zip(*res):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    assert np.abs(res[0][0]).max() > 0.1 and not res[0][2].any() and not res[0][3].any()
