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
    return res


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
