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
