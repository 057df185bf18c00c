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
from slab_oracle_compute import OracleChorin, OracleCompute, OracleSor

NX, NY, B = 16, 24, 2
DT, RHO, NU, LX, LY = 1e-2, 1.3, 0.05, 2 * np.pi, 3.0


def fields(nx=NX, ny=NY):
    rng = np.random.default_rng(42)
    return [rng.standard_normal((B, nx, ny)) for _ in range(5)]


def _worker(rank, world, port, out, NX=NX, NY=NY):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nns.slab import SlabResidual
        f = fields(NX, NY)
        nloc = NX // world
        loc = [torch.from_numpy(np.ascontiguousarray(a[:, rank * nloc:(rank + 1) * nloc])) for a in f]
        s = SlabResidual(NX, NY, DT, RHO, NU, LX, LY, compute=OracleCompute())
        # halo messages really are the periodic neighbours' edge rows
        h, top, bot = s.start_halo(loc[:3])
        h.wait()
        for k in range(3):
            np.testing.assert_array_equal(top[k].numpy(), f[k][:, (rank * nloc - 1) % NX])
            np.testing.assert_array_equal(bot[k].numpy(), f[k][:, ((rank + 1) * nloc) % NX])
        if world == 2:
            # two ranks: both ring neighbours are the same peer.  SlabResidual's halo buffers are adjacent pairs, so ONE message each way carries
            # both rows; separate buffers go as two ordered messages -- same rows either way
            posted = []
            orig = s.tr.sendrecv
            s.tr.sendrecv = lambda sends, recvs: (posted.append((len(sends), len(recvs))), orig(sends, recvs))[1]
            s.start_halo(loc[:3])[0].wait()
            sep = [torch.empty(3, B, NY, dtype=torch.float64) for _ in range(4)]
            sep[0].copy_(torch.stack([t[:, 0] for t in loc[:3]])), sep[1].copy_(torch.stack([t[:, -1] for t in loc[:3]]))
            s.tr.ring_exchange(sep[0], sep[1], sep[2], sep[3], wrap=True).wait()
            s.tr.sendrecv = orig
            assert posted == [(1, 1), (2, 2)], posted
            assert torch.equal(sep[2], bot) and torch.equal(sep[3], top)
        # the all-to-all delivers, from every source rank, its rows of this rank's column block; the return trip inverts it
        nyl = NY // world
        send, recv = torch.empty(world, 3, B, nloc, nyl, dtype=torch.float64), torch.empty(world, 3, B, nloc, nyl, dtype=torch.float64)
        s.compute.transpose_pack(loc[:3], send, world)
        s.tr.all_to_all(recv, send).wait()
        for src in range(world):
            for k in range(3):
                np.testing.assert_array_equal(recv[src, k].numpy(), f[k][:, src * nloc:(src + 1) * nloc, rank * nyl:(rank + 1) * nyl])
        back = torch.empty_like(recv)
        s.tr.all_to_all(back, recv).wait()
        got = [torch.empty_like(t) for t in loc[:3]]
        s.compute.transpose_unpack(back, got, world)
        for k in range(3):
            np.testing.assert_array_equal(got[k].numpy(), loc[k].numpy())
        res = {}
        for st in (5, 9):
            res['fd%d' % st] = [t.numpy() for t in s.fd(*loc, stencil=st)]
        res['spec'] = [t.numpy() for t in s.spectral(*loc)]
        both = s.both(*loc)                                     # the fused form: halo under the transposes, one row pass
        res['bfd'], res['bspec'] = [t.numpy() for t in both[0]], [t.numpy() for t in both[1]]
        # the batch-chunk pipeline (what a stream-ordered transport runs by default): same numbers whatever the chunking
        both2 = s.both(*loc, chunks=2)
        spec2 = s.spectral(*loc, chunks=2)
        for a, b in zip(list(both2[0]) + list(both2[1]) + list(spec2), res['bfd'] + res['bspec'] + res['spec']):
            np.testing.assert_array_equal(a.numpy(), b)
        np.savez(os.path.join(out, 'r%d.npz' % rank), **{k + '_%d' % i: a for k, v in res.items() for i, a in enumerate(v)})
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world,NX,NY', [(2, NX, NY), (4, NX, NY), (8, 32, 40)])
def test_slab_decomposition_matches_single_process(world, NX, NY, tmp_path):
    """(world = 8: the node's rank count -- 4 rows and 5 columns per rank; every rank has two distinct ring neighbours and seven all-to-all peers.)"""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), NX, NY), nprocs=world, join=True)
    f = fields(NX, NY)
    hx, hy = LX / NX, LY / NY
    ref = {'fd5': OP.fd_residual(*f, DT, hx, hy, RHO, NU, 5), 'fd9': OP.fd_residual(*f, DT, hx, hy, RHO, NU, 9),
           'spec': OP.spectral_residual(*f, DT, LX, LY, RHO, NU)}
    ref['bfd'], ref['bspec'] = ref['fd5'], ref['spec']
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
        # early stop in the middle of an enqueued chunk (check_every = 5): the sweeps after the stopping one switch themselves off
        s2 = SlabPressure(PNX, PNY, 1.0 / PNX, 1.0 / PNY, PBETA, tol=0.5, compute=OracleSor(), check_every=5)
        pl2 = torch.from_numpy(np.array(s2.local_rows(pressure_problem()[0])))          # a fresh copy: with one rank the first solve ran in place on p0
        done2, err2 = s2.solve_(pl2, torch.from_numpy(np.ascontiguousarray(s2.local_rows(C))), PCAP)
        np.save(os.path.join(out, 'q%d.npy' % rank), pl2.numpy())
        np.save(os.path.join(out, 'j%d.npy' % rank), np.array([done2, err2]))
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
    ref = p0.copy(); err, sweeps, prev = 1.0, 0, p0.copy()
    while err > 0.5 and sweeps < PCAP:
        O.sor_sweep_redblack(ref, C, 1.0 / PNX, 1.0 / PNY, PBETA)
        err = np.max(np.abs(ref - prev)); prev = ref.copy(); sweeps += 1
    assert 0 < sweeps < PCAP and sweeps % 5 != 0                           # the stop falls inside a chunk of 5 enqueued sweeps
    got = np.concatenate([np.load(os.path.join(str(tmp_path), 'q%d.npy' % r)) for r in range(world)])
    assert np.array_equal(got, ref)
    assert all(int(j[0]) == sweeps and j[1] == err for j in (np.load(os.path.join(str(tmp_path), 'j%d.npy' % r)) for r in range(world)))


# ------------------------------------------------------------------------------------------------------------------
# the whole chorin_fd cavity step sharded by rows (nns.slab.SlabChorinFD): bitwise the single-process oracle run
# ------------------------------------------------------------------------------------------------------------------
CNX, CNY, CNT, CNIT = 22, 17, 6, 30


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


def _loopback_worker(rank, world, port, out):
    """loopback=True is the one-GPU rehearsal of the RCCL transport (a rank's messages go through the process group to itself,
    tests/test_gpu_multirank.py); gloo has no self-send, so on a world-1 gloo group the request degrades to the local copies."""
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nns.slab import SlabResidual
        f = fields()
        loc = [torch.from_numpy(np.ascontiguousarray(a)) for a in f]
        s = SlabResidual(NX, NY, DT, RHO, NU, LX, LY, compute=OracleCompute(), loopback=True)
        assert not s.tr.loopback and s.tr.local
        ref = OP.spectral_residual(*f, DT, LX, LY, RHO, NU)
        for chunks in (1, 2):
            for x, y in zip(s.both(*loc, chunks=chunks)[1], ref):
                np.testing.assert_allclose(x.numpy(), y, rtol=1e-10, atol=1e-10)
    finally:
        dist.destroy_process_group()


def test_world_size_one_loopback_request_on_gloo_stays_local(tmp_path):
    mp.spawn(_loopback_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
