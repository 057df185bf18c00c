"""The multi-rank paths on the HIP back-end (BASELINE configs 4 and 5): two ranks started as fresh child
processes, both on cuda:0 of a one-GPU box, transport gloo (nns/_comm.py stages device buffers through the host there;
on a GPU node the same code runs on RCCL), compute = HipCompute / HipChorinCompute.  Results are compared with the
single-process HIP operators in THIS process:

  * SlabResidual.fd (5-, 9-point) / spectral / both at 1024^2 (cfg 4 shape), B = 2: bitwise equal to ops.fd_residual,
    ops.spec_residual, ops.residual_both on the whole grids, and <= 1e-5 rel-L2 from the float64 oracle;
  * SlabChorinFD explicit + semi_implicit (float64): bitwise the single-process NavierStokesSystem(pressure_solver='redblack');
  * ensemble training (cfg 5 shape, members sharded): ONE flat all-reduce carrying the gradient bucket and the local sum of
    squares reproduces the single-process full-batch gradient of || pred - obs ||_2, replicas stay bit-identical."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_l2
import mr_cases as MC

pytestmark = pytest.mark.gpu
WORLD = 2


def run_ranks(case, out, world=WORLD, backend="gloo", device="0", timeout=600):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   MR_BACKEND=backend, MR_DEVICE=device)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'mr_worker.py'), case, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, logs[r][-3000:])
    return [np.load(os.path.join(out, '%s_r%d.npz' % (case, r))) for r in range(world)]


def test_rccl_world1_loopback_slab_paths_bitwise(tmp_path, gpu_device):
    """The RCCL branch of the transport on the hardware a one-GPU box offers: ONE rank whose messages go through the RCCL process
    group to itself (tests/mr_cases.py rank_loopback)."""
    parts = run_ranks("loopback", str(tmp_path), world=1, backend="nccl", timeout=300)
    assert int(parts[0]['checks']) == 8


def _check_slab_residual(parts):
    from nns import ops
    from oracle import periodic as OP
    f = MC.residual_fields()
    d = [torch.as_tensor(a, device='cuda') for a in f]
    h = MC.L / MC.N
    cat = lambda key: [np.concatenate([p['%s_%d' % (key, i)] for p in parts], axis=1) for i in range(3)]
    single = {'fd5': ops.fd_residual(*d, MC.DT, h, h, MC.RHO, MC.NU, 5), 'fd9': ops.fd_residual(*d, MC.DT, h, h, MC.RHO, MC.NU, 9),
              'spec': ops.spec_residual(*d, MC.DT, MC.L, MC.L, MC.RHO, MC.NU)}
    single['bfd'], single['bspec'] = ops.residual_both(*d, MC.DT, MC.L, MC.L, MC.RHO, MC.NU)
    single['b2fd'], single['b2spec'] = ops.residual_both(*d, MC.DT, MC.L, MC.L, MC.RHO, MC.NU, precise=2)
    f64 = [a.astype(np.float64) for a in f]
    oracle = {'fd5': OP.fd_residual(*f64, MC.DT, h, h, MC.RHO, MC.NU, 5), 'fd9': OP.fd_residual(*f64, MC.DT, h, h, MC.RHO, MC.NU, 9),
              'spec': OP.spectral_residual(*f64, MC.DT, MC.L, MC.L, MC.RHO, MC.NU)}
    oracle['bfd'], oracle['bspec'] = oracle['fd5'], oracle['spec']
    oracle['b2fd'], oracle['b2spec'] = oracle['fd5'], oracle['spec']
    for key in ('fd5', 'fd9', 'spec', 'bfd', 'bspec', 'b2fd', 'b2spec'):
        got = cat(key)
        for i in range(3):
            assert got[i].shape == (MC.B, MC.N, MC.N)
            assert rel_l2(got[i], oracle[key][i]) <= 1e-5, (key, i)
            assert np.array_equal(got[i], single[key][i].cpu().numpy()), '%s[%d]: two-rank slab result differs from the single-process kernel' % (key, i)


@pytest.mark.parametrize('world', [2, 4])
def test_slab_residual_two_ranks_hip_1024(world, tmp_path, gpu_device):
    """(world = 4: each rank has TWO distinct ring neighbours and three all-to-all peers; at most 6 processes may share the card.)"""
    _check_slab_residual(run_ranks('residual', str(tmp_path), world))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per GPU (a one-GPU box runs the loopback and gloo forms)")
def test_slab_residual_two_ranks_rccl_1024(tmp_path, gpu_device):
    """The same checks with the transport on RCCL, one rank per GPU (device buffers, grouped send/recv, async all-to-all with
    stream-ordered waits, the batch-chunk pipeline; P = 2: both ring neighbours are the same peer)."""
    _check_slab_residual(run_ranks('residual', str(tmp_path), 2, backend='nccl', device='rank'))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per GPU")
def test_slab_chorin_two_ranks_rccl_bitwise(tmp_path, gpu_device):
    """SlabChorinFD / SlabPressure on RCCL: halo exchange per half-sweep, int-view all-reduce(MAX) of the error slots."""
    _check_slab_chorin(run_ranks('chorin96', str(tmp_path), 2, backend='nccl', device='rank'), 96)


@pytest.mark.parametrize('world', [2, 3])
def test_slab_chorin_two_ranks_hip_bitwise(world, tmp_path, gpu_device):
    """(world = 3: uneven 96 = 32 + 32 + 32 rows is even, so the cavity is 98 wide there: 33 + 33 + 32, and the middle rank owns no wall.)"""
    n = 96 if world == 2 else 98
    _check_slab_chorin(run_ranks('chorin%d' % n, str(tmp_path), world), n)


def _check_slab_chorin(parts, n):
    from nns.chorin_fd import NavierStokesSystem
    MC.CN = n
    ics, (u_bc, v_bc, p_bc) = MC.cavity_problem()
    for method, axis in (('explicit', 1), ('semi_implicit', 2)):
        s = NavierStokesSystem(*[a.copy() for a in ics], u_bc, v_bc, p_bc, nt=MC.CNT, nit=MC.CNIT, nx=MC.CN, ny=MC.CN, dt=1e-3, rho=1.0, nu=0.05,
                               beta=1.25, method=method, pressure_solver='redblack')
        ref = s.simulate()
        for name, r in zip('uvp', ref):
            got = np.concatenate([p['%s_%s' % (method, name)] for p in parts], axis=axis)
            assert got.shape == r.shape and np.array_equal(got, r), (method, name)
        assert all(np.array_equal(parts[0][method + '_sor'], q[method + '_sor']) for q in parts[1:])          # every rank saw the same sweep count and err
        assert np.abs(ref[0][-1]).max() > 1e-3


def test_ensemble_two_ranks_hip_flat_allreduce(tmp_path, gpu_device):
    parts = run_ranks('ensemble', str(tmp_path))
    m = MC.ensemble_model()
    obs = MC.ensemble_obs()
    t = torch.arange(MC.ENT, device='cuda') + 1
    loss = m.loss(obs[0], t, obs)                               # single process, the whole ensemble, the reference's objective
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    g0 = parts[0]['g0']
    assert np.array_equal(g0, parts[1]['g0'])                    # one all-reduce: both ranks hold the same bucket
    assert rel_l2(g0, ref) < 2e-5                                 # float32 sums in a different order (2 shards vs 1 sweep)
    assert abs(parts[0]['losses'][0] - float(loss.detach())) <= 1e-5 * float(loss.detach())
    assert np.array_equal(parts[0]['params'], parts[1]['params'])                # replicas identical after 3 Adam steps
    assert parts[0]['losses'][-1] < parts[0]['losses'][0]
