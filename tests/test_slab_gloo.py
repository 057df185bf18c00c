"""Multi-process (gloo, CPU) tests of the slab decomposition (nns/slab.py): halo exchange, packed
all-to-all transposes and their inverses, for world sizes 2 and 4.  The communication logic is what
is under test; the per-slab compute is injected from the CPU oracle (the product's compute back-end is
HIP-only), and the assembled result must equal the single-process oracle on the full grid."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import periodic as OP

NX, NY, B = 16, 24, 2
DT, RHO, NU, LX, LY = 1e-2, 1.3, 0.05, 2 * np.pi, 3.0


class OracleCompute(object):
    """CPU stand-in for nns.slab.HipCompute (tests only)."""

    def fd_residual(self, u, v, p, up, vp, dt, dx, dy, rho, nu, stencil):
        r = OP.fd_residual(*[t.numpy() for t in (u, v, p, up, vp)], dt, dx, dy, rho, nu, stencil)
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in r)

    def spec_xpass(self, u, v, p, Lx, rho, nu, precise):
        r = OP.spectral_xpart(u.numpy(), v.numpy(), p.numpy(), Lx, rho, nu)
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in r)

    def spec_ypass(self, u, v, p, up, vp, ru, rv, rd, dt, Ly, rho, nu, precise):
        r = OP.spectral_ypart(*[t.numpy() for t in (u, v, p, up, vp, ru, rv, rd)], dt, Ly, rho, nu)
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in r)


def fields():
    rng = np.random.default_rng(42)
    return [rng.standard_normal((B, NX, NY)) for _ in range(5)]


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nns.slab import SlabResidual
        f = fields()
        nloc = NX // world
        loc = [torch.from_numpy(np.ascontiguousarray(a[:, rank * nloc:(rank + 1) * nloc])) for a in f]
        s = SlabResidual(NX, NY, DT, RHO, NU, LX, LY, compute=OracleCompute())
        # halo rows really are the periodic neighbours' edge rows
        padded = s.exchange_halo(loc[:3])
        for k in range(3):
            np.testing.assert_array_equal(padded[k][:, 0].numpy(), f[k][:, (rank * nloc - 1) % NX])
            np.testing.assert_array_equal(padded[k][:, -1].numpy(), f[k][:, ((rank + 1) * nloc) % NX])
        # transposes are exact inverses and put the right data in the right place
        cols = s._to_columns(loc[:3])
        nyl = NY // world
        for k in range(3):
            np.testing.assert_array_equal(cols[k].numpy(), f[k][:, :, rank * nyl:(rank + 1) * nyl])
        back = s._to_rows(cols)
        for k in range(3):
            np.testing.assert_array_equal(back[k].numpy(), loc[k].numpy())
        res = {}
        for st in (5, 9):
            res['fd%d' % st] = [t.numpy() for t in s.fd(*loc, stencil=st)]
        res['spec'] = [t.numpy() for t in s.spectral(*loc)]
        np.savez(os.path.join(out, 'r%d.npz' % rank), **{k + '_%d' % i: a for k, v in res.items() for i, a in enumerate(v)})
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world', [2, 4])
def test_slab_decomposition_matches_single_process(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    f = fields()
    hx, hy = LX / NX, LY / NY
    ref = {'fd5': OP.fd_residual(*f, DT, hx, hy, RHO, NU, 5), 'fd9': OP.fd_residual(*f, DT, hx, hy, RHO, NU, 9),
           'spec': OP.spectral_residual(*f, DT, LX, LY, RHO, NU)}
    parts = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    for key, r in ref.items():
        for i in range(3):
            got = np.concatenate([p['%s_%d' % (key, i)] for p in parts], axis=1)
            np.testing.assert_allclose(got, r[i], rtol=1e-10, atol=1e-10)


def test_split_oracle_passes_equal_full_oracle():
    f = fields()
    pu, pv, pd = OP.spectral_xpart(f[0], f[1], f[2], LX, RHO, NU)
    got = OP.spectral_ypart(*f, pu, pv, pd, DT, LY, RHO, NU)
    ref = OP.spectral_residual(*f, DT, LX, LY, RHO, NU)
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11)


def test_world_size_one_degenerates_to_local(tmp_path):
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)


# ------------------------------------------------------------------------------------------------------------------
# sharded red-black pressure solve (nns.slab.SlabPressure): halo exchange per half-sweep + all-reduce(max) per sweep
# ------------------------------------------------------------------------------------------------------------------
PNX, PNY, PBETA, PCAP = 27, 14, 1.5, 60


class OracleSor(object):
    """CPU stand-in for HipSorCompute (tests only): one colour of oracle.chorin_fd.sor_sweep_redblack on the slab."""

    def halfsweep(self, p, C, err, gi0, colour, dx, dy, beta):
        a = p.numpy()
        nxl, ny = a.shape
        I, J = np.meshgrid(np.arange(1, nxl - 1), np.arange(1, ny - 1), indexing='ij')
        m = ((I + gi0 + J) % 2) == colour
        i, j = I[m], J[m]
        c = C.numpy()
        new = (beta * (dy**2 * a[i + 1, j] + dy**2 * a[i - 1, j] + dx**2 * a[i, j + 1] + dx**2 * a[i, j - 1] - c[i, j]) / (2 * dx**2 + 2 * dy**2)
               + (1 - beta) * a[i, j])
        if new.size:
            err[0] = max(float(err[0]), float(np.max(np.abs(new - a[i, j]))))
        a[i, j] = new
        return err

    def err_value(self, err):
        return err


def pressure_problem():
    rng = np.random.default_rng(5)
    return 0.01 * rng.standard_normal((PNX, PNY)), 0.1 * rng.standard_normal((PNX, PNY))


def _pworker(rank, world, port, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nns.slab import SlabPressure
        p0, C = pressure_problem()
        s = SlabPressure(PNX, PNY, 1.0 / PNX, 1.0 / PNY, PBETA, tol=1e-9, compute=OracleSor())
        pl = torch.from_numpy(np.ascontiguousarray(s.local_rows(p0)))
        done, err = s.solve_(pl, torch.from_numpy(np.ascontiguousarray(s.local_rows(C))), PCAP)
        np.save(os.path.join(out, 'p%d.npy' % rank), pl.numpy())
        np.save(os.path.join(out, 'i%d.npy' % rank), np.array([done, err, s.lo, s.hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2, 4])
def test_slab_pressure_redblack(world, tmp_path):
    """The sharded red-black solve is bitwise the single-process one (uneven row split 27 = 7+7+7+6, early stop off)."""
    from oracle import chorin_fd as O
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_pworker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p0, C = pressure_problem()
    ref = p0.copy(); err, sweeps, prev = 1.0, 0, p0.copy()
    while err > 1e-9 and sweeps < PCAP:
        O.sor_sweep_redblack(ref, C, 1.0 / PNX, 1.0 / PNY, PBETA)
        err = np.max(np.abs(ref - prev)); prev = ref.copy(); sweeps += 1
    got = np.concatenate([np.load(os.path.join(str(tmp_path), 'p%d.npy' % r)) for r in range(world)])
    info = [np.load(os.path.join(str(tmp_path), 'i%d.npy' % r)) for r in range(world)]
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert all(int(i[0]) == sweeps and i[1] == err for i in info)
    assert [int(i[2]) for i in info] == [sum(len(x) for x in np.array_split(np.arange(PNX), world)[:r]) for r in range(world)]


# ------------------------------------------------------------------------------------------------------------------
# the whole chorin_fd cavity step sharded by rows (nns.slab.SlabChorinFD): bitwise the single-process oracle run
# ------------------------------------------------------------------------------------------------------------------
CNX, CNY, CNT, CNIT = 22, 17, 6, 30


class OracleChorin(OracleSor):
    """CPU stand-in for HipChorinCompute (tests only): the oracle's operators applied to the slab arrays."""

    def predictor(self, un, vn, un1, vn1, dt, dx, dy, nu, corrected):
        from oracle import chorin_fd as O
        f = O.explicit_predictor_corrected if corrected else O.explicit_predictor
        ui, vi = f(un.numpy(), vn.numpy(), un1.numpy(), vn1.numpy(), dt, dx, dy, nu)
        return torch.from_numpy(ui), torch.from_numpy(vi)

    def predictor_adi(self, un, vn, un1, vn1, dt, dx, dy, nu):
        from oracle import chorin_fd as O
        ui, vi = O.semi_implicit_predictor(un.numpy(), vn.numpy(), un1.numpy(), vn1.numpy(), dt, dx, dy, nu, column_slab=True)
        return torch.from_numpy(ui), torch.from_numpy(vi)

    def bc_apply_(self, A, bcs):
        from oracle.boundary import apply_bc_list
        apply_bc_list(A.numpy(), bcs)
        return A

    def rhs(self, ui, vi, dt, dx, dy, rho):
        from oracle import chorin_fd as O
        return torch.from_numpy(O.pressure_rhs(ui.numpy(), vi.numpy(), dt, dx, dy, rho))

    def correction(self, ui, vi, p, dt, dx, dy):
        from oracle import chorin_fd as O
        u, v = O.correction(ui.numpy(), vi.numpy(), p.numpy(), dt, dx, dy)
        return torch.from_numpy(u), torch.from_numpy(v)


def cavity_problem():
    from oracle.boundary import cavity_bcs
    rng = np.random.default_rng(9)
    dx, dy = 2. / (CNX - 1), 2. / (CNY - 1)
    ics = [0.05 * rng.standard_normal((CNX, CNY)) for _ in range(3)]        # non-trivial ICs: the BCs must overwrite the edges
    return ics, cavity_bcs(dx, dy)


def _cworker(rank, world, port, out, advection):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nns.slab import SlabChorinFD
        ics, (u_bc, v_bc, p_bc) = cavity_problem()
        s = SlabChorinFD(u_bc, v_bc, p_bc, CNIT, CNX, CNY, 1e-3, 1.0, 0.05, 1.25, advection=advection, compute=OracleChorin())
        us, vs, ps = s.simulate(*[torch.from_numpy(a.copy()) for a in ics], CNT)
        np.savez(os.path.join(out, 'c%d.npz' % rank), u=us.numpy(), v=vs.numpy(), p=ps.numpy(), sor=np.array(s.last_sor))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,advection', [(1, 'reference'), (2, 'reference'), (4, 'corrected'), (3, 'reference')])
def test_slab_chorin_fd_cavity(world, advection, tmp_path):
    """Row-sharded cavity run == oracle.simulate(pressure_solver='redblack') on the whole grid, bitwise (float64): the
    lid on 'right' lives on the last rank only, Neumann 'left' / 'right' pressure rows on the edge ranks, uneven splits
    (22 rows over 3 and 4 ranks)."""
    from oracle import chorin_fd as O
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_cworker, args=(world, port, str(tmp_path), advection), nprocs=world, join=True)
    ics, (u_bc, v_bc, p_bc) = cavity_problem()
    ur, vr, pr = O.simulate(*[a.copy() for a in ics], u_bc, v_bc, p_bc, CNT, CNIT, 1e-3, 1.0, 0.05, 1.25, 'explicit',
                            advection=advection, pressure_solver='redblack')
    parts = [np.load(os.path.join(str(tmp_path), 'c%d.npz' % r)) for r in range(world)]
    for name, ref in (('u', ur), ('v', vr), ('p', pr)):
        got = np.concatenate([d[name] for d in parts], axis=1)
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    assert len({tuple(d['sor']) for d in parts}) == 1                      # every rank saw the same sweep count and err
    assert np.abs(ur[-1]).max() > 1e-3                                      # the lid drives a flow


SN = 21          # the reference's ADI needs a square grid


def _aworker(rank, world, port, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nns.slab import SlabChorinFD
        from oracle.boundary import cavity_bcs
        rng = np.random.default_rng(4)
        ics = [0.05 * rng.standard_normal((SN, SN)) for _ in range(3)]
        h = 2. / (SN - 1)
        u_bc, v_bc, p_bc = cavity_bcs(h, h)
        s = SlabChorinFD(u_bc, v_bc, p_bc, CNIT, SN, SN, 1e-3, 1.0, 0.05, 1.25, method='semi_implicit', compute=OracleChorin())
        us, vs, ps = s.simulate(*[torch.from_numpy(a.copy()) for a in ics], CNT)
        np.savez(os.path.join(out, 'a%d.npz' % rank), u=us.numpy(), v=vs.numpy(), p=ps.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2, 4])
def test_slab_chorin_fd_semi_implicit_column_slabs(world, tmp_path):
    """The reference's ADI step on COLUMN slabs (both tridiagonal solves run along axis 0: no communication for them;
    'bottom' lives on the first rank, 'top' on the last) == oracle.simulate(method='semi_implicit',
    pressure_solver='redblack') on the whole grid, bitwise."""
    from oracle import chorin_fd as O
    from oracle.boundary import cavity_bcs
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_aworker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(4)
    ics = [0.05 * rng.standard_normal((SN, SN)) for _ in range(3)]
    h = 2. / (SN - 1)
    u_bc, v_bc, p_bc = cavity_bcs(h, h)
    ur, vr, pr = O.simulate(*[a.copy() for a in ics], u_bc, v_bc, p_bc, CNT, CNIT, 1e-3, 1.0, 0.05, 1.25, 'semi_implicit',
                            pressure_solver='redblack')
    parts = [np.load(os.path.join(str(tmp_path), 'a%d.npz' % r)) for r in range(world)]
    for name, ref in (('u', ur), ('v', vr), ('p', pr)):
        got = np.concatenate([d[name] for d in parts], axis=2)
        assert got.shape == ref.shape and np.array_equal(got, ref), name
