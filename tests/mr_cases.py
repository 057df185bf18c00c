"""Problem definitions shared by the ranks (tests/mr_worker.py) and the single-process side (tests/test_gpu_multirank.py)
of the multi-rank GPU tests.  Everything here runs on the HIP back-end."""
import numpy as np
import torch

# ---- cfg 4 shape: 1024 x 1024 periodic box, slab-decomposed by rows -------------------------------------------------
N, B = 1024, 2
DT, RHO, NU, L = 1e-3, 1.0, 2 * np.pi / 1000, 2 * np.pi


def residual_fields():
    from nns.synthetic import residual_inputs
    return residual_inputs(B, N, dt=DT, nu=NU, rho=RHO)


def rank_residual(rank, world):
    from nns.slab import SlabResidual
    f = residual_fields()
    nloc = N // world
    loc = [torch.as_tensor(np.ascontiguousarray(a[:, rank * nloc:(rank + 1) * nloc]), device='cuda') for a in f]
    s = SlabResidual(N, N, DT, RHO, NU, L, L)
    res = {}
    for st in (5, 9):
        for i, t in enumerate(s.fd(*loc, stencil=st)):
            res['fd%d_%d' % (st, i)] = t.cpu().numpy()
    for i, t in enumerate(s.spectral(*loc)):
        res['spec_%d' % i] = t.cpu().numpy()
    bf, bs = s.both(*loc)
    for i in range(3):
        res['bfd_%d' % i], res['bspec_%d' % i] = bf[i].cpu().numpy(), bs[i].cpu().numpy()
    # the float64-forward arithmetic of the same paths (segmented column pass and halo row pass with TF = double)
    s2 = SlabResidual(N, N, DT, RHO, NU, L, L, precise=2)
    b2f, b2s = s2.both(*loc)
    for i in range(3):
        res['b2fd_%d' % i], res['b2spec_%d' % i] = b2f[i].cpu().numpy(), b2s[i].cpu().numpy()
    # back-to-back calls reuse the message buffers; the batch-chunk pipeline (the default on RCCL) must not change a bit
    for chunks in (1, 2, 2):
        bf2, bs2 = s.both(*loc, chunks=chunks)
        sp2 = s.spectral(*loc, chunks=chunks)
        for i in range(3):
            assert torch.equal(bf2[i], bf[i]) and torch.equal(bs2[i], bs[i]), ('both', chunks, i)
            assert np.array_equal(sp2[i].cpu().numpy(), res['spec_%d' % i]), ('spectral', chunks, i)
    ofd, osp = tuple(torch.empty_like(loc[0]) for _ in range(3)), tuple(torch.empty_like(loc[0]) for _ in range(3))
    for rep in range(3):                                     # recorded, then replayed twice (caller-owned outputs)
        s.both(*loc, chunks=2, out_fd=ofd, out_spec=osp)
        for i in range(3):
            assert torch.equal(ofd[i], bf[i]) and torch.equal(osp[i], bs[i]), ('both, replayed', rep, i)
    return res


def rank_loopback(rank, world):
    """World-1 RCCL LOOPBACK (nns/_comm.py): every message of the slab paths goes through the RCCL process group to this rank
    itself -- grouped send/recv on device buffers with both ring neighbours the same peer, async all_to_all_single, stream-ordered
    wait(), the integer-view all-reduce(MAX) -- and the results must be bitwise the single-process kernels'."""
    import torch.distributed as dist
    from nns import ops
    from nns.slab import SlabResidual
    from nns._comm import Transport
    assert world == 1 and dist.get_backend() == 'nccl'
    f = residual_fields()
    d = [torch.as_tensor(a, device='cuda') for a in f]
    s = SlabResidual(N, N, DT, RHO, NU, L, L, loopback=True)
    assert s.tr.loopback and not s.tr.local
    h = L / N
    ref = {5: ops.fd_residual(*d, DT, h, h, RHO, NU, 5), 9: ops.fd_residual(*d, DT, h, h, RHO, NU, 9)}
    ref_sp = ops.spec_residual(*d, DT, L, L, RHO, NU)
    ref_bfd, ref_bsp = ops.residual_both(*d, DT, L, L, RHO, NU)
    checks = 0
    for rep in range(3):                                        # back to back: the message buffers are reused
        for st in (5, 9):
            for a, b in zip(s.fd(*d, stencil=st), ref[st]):
                assert torch.equal(a, b), ('fd', st, rep)
        for chunks in (1, 2):
            for a, b in zip(s.spectral(*d, chunks=chunks), ref_sp):
                assert torch.equal(a, b), ('spectral', chunks, rep)
            bf, bs = s.both(*d, chunks=chunks)
            for a, b in zip(list(bf) + list(bs), list(ref_bfd) + list(ref_bsp)):
                assert torch.equal(a, b), ('both', chunks, rep)
            checks += 1
    # caller-owned outputs: the first evaluation records its C calls, the next ones replay them (SlabResidual._both_replay)
    for chunks in (1, 2):
        ofd, osp = tuple(torch.empty_like(d[0]) for _ in range(3)), tuple(torch.empty_like(d[0]) for _ in range(3))
        for rep in range(3):
            for t in ofd + osp:
                t.fill_(float('nan'))
            bf, bs = s.both(*d, chunks=chunks, out_fd=ofd, out_spec=osp)
            assert bf is ofd and bs is osp and len(s._plans) >= 1
            for a, b in zip(list(bf) + list(bs), list(ref_bfd) + list(ref_bsp)):
                assert torch.equal(a, b), ('both, replayed', chunks, rep)
        checks += 1
    # the SOR error slots travel as an integer view through all-reduce(MAX) (nns/slab.py SlabPressure.solve_slab_)
    tr = Transport(loopback=True)
    slots = torch.tensor([1.0, 0.25, float('nan'), 0.0], device='cuda')
    before = slots.clone()
    tr.all_reduce_(slots.view(torch.int32)[1:3], dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    assert torch.equal(slots.view(torch.int32), before.view(torch.int32))
    return dict(checks=np.array(checks))


# ---- the chorin_fd cavity step sharded over ranks (float64: bitwise) -----------------------------------------------
CN, CNT, CNIT = 96, 4, 50


def cavity_problem():
    from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as Nm
    h = 2. / (CN - 1)
    rng = np.random.default_rng(11)
    ics = [0.05 * rng.standard_normal((CN, CN)) for _ in range(3)]
    u_bc = [D(0., 'left', h, h), D(1., 'right', h, h), D(0., 'bottom', h, h), D(0., 'top', h, h)]
    v_bc = [D(0., 'left', h, h), D(0., 'right', h, h), D(0., 'bottom', h, h), D(0., 'top', h, h)]
    p_bc = [Nm(0., 'left', h, h), Nm(0., 'right', h, h), Nm(0., 'bottom', h, h), D(0., 'top', h, h)]
    return ics, (u_bc, v_bc, p_bc)


def rank_chorin96(rank, world):
    return rank_chorin(rank, world, 96)


def rank_chorin98(rank, world):
    return rank_chorin(rank, world, 98)


def rank_chorin(rank, world, n=None):
    global CN
    if n is not None:
        CN = n
    from nns.slab import SlabChorinFD
    ics, (u_bc, v_bc, p_bc) = cavity_problem()
    res = {}
    for method in ('explicit', 'semi_implicit'):
        s = SlabChorinFD(u_bc, v_bc, p_bc, CNIT, CN, CN, 1e-3, 1.0, 0.05, 1.25, method=method)
        us, vs, ps = s.simulate(*[torch.as_tensor(a.copy(), device='cuda') for a in ics], CNT)
        res.update({method + '_u': us.cpu().numpy(), method + '_v': vs.cpu().numpy(), method + '_p': ps.cpu().numpy(),
                    method + '_sor': np.array(s.last_sor, dtype=np.float64)})
    return res


# ---- cfg 5 shape (reduced member count): ensemble sharded over ranks, ONE flat all-reduce ----------------------------
EK, EN, ENT, EMB = 10, 256, 4, 8            # K = 10 coefficients, 256 x 256 fields (cfg 5), nt = 4, 8 members in total


def ensemble_model():
    from nns.neural_spectral.spectral_ode import PDEFunc
    torch.manual_seed(5)
    return PDEFunc(EK, EN, EN).cuda()


def ensemble_obs():
    g = torch.Generator().manual_seed(17)
    return torch.randn(ENT, EMB, 3, EN, EN, generator=g).cuda()


def rank_ensemble(rank, world):
    from nns.data_parallel import FlatGradAllReduce, broadcast_parameters, norm_loss_step
    m = ensemble_model()
    if rank == 1:
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)                                  # deliberately different start: the broadcast must fix it
    broadcast_parameters(list(m.parameters()))
    obs = ensemble_obs()
    per = EMB // world
    shard = obs[:, rank * per:(rank + 1) * per].contiguous()
    t = torch.arange(ENT, device='cuda') + 1
    bucket = FlatGradAllReduce(m.parameters(), average=False, extra=1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses, g0 = [], None
    for step in range(3):
        loss = norm_loss_step(m, bucket, opt, lambda: m.sumsq(shard[0], t, shard))
        losses.append(float(loss))
        if step == 0:
            g0 = bucket.flat[:-1].detach().cpu().numpy().copy()
    return dict(g0=g0, losses=np.array(losses), params=torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy())
