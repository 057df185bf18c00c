"""Developer helper (GPU box): random cases through round 4's chorin_fd paths against the forms they replace, BITWISE:
  1. the LDS-resident semi-implicit predictor vs the streaming kernel (NNS_ADI_LDS=0 in a child process) -- random sizes, batches, dtypes, advection forms
  2. the one-launch explicit step vs the separate operators -- random boundary lists (kinds, sides, values, order, repeats), sizes, batches, sweep caps
  3. the lexicographic SOR with random sweep-count hints vs none
usage: fd_fuzz.py SEED N   (prints one line per case, exit status 1 if any differ)"""
import os, sys, subprocess, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
seed, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
rng = np.random.default_rng(seed)

def adi_cases():
    out = []
    for _ in range(N):
        corrected = bool(rng.integers(0, 2))
        nx = int(rng.integers(5, 97)); ny = int(rng.integers(5, 97)) if corrected else nx
        out.append(dict(nx=nx, ny=ny, B=int(rng.integers(1, 4)), f64=bool(rng.integers(0, 2)), corrected=corrected, dt=float(rng.choice([1e-3, 1e-2])),
                        nu=float(rng.choice([0.02, 0.5])), dx=float(rng.uniform(0.01, 0.1)), dy=float(rng.uniform(0.01, 0.1)), seed=int(rng.integers(1 << 30))))
    return out

if len(sys.argv) > 3 and sys.argv[3] == '--adi-child':
    import torch
    from nns import ops
    cases = json.load(open(sys.argv[4])); outs = {}
    for k, c in enumerate(cases):
        r = np.random.default_rng(c['seed']); dt_ = torch.float64 if c['f64'] else torch.float32
        f = [torch.as_tensor(r.standard_normal((c['B'], c['nx'], c['ny'])), dtype=dt_, device='cuda') for _ in range(4)]
        ui, vi = ops.fd_predictor_adi(*f, c['dt'], c['dx'], c['dy'] if c['corrected'] else c['dx'], c['nu'], corrected=c['corrected'])
        outs['u%d' % k], outs['v%d' % k] = ui.cpu().numpy(), vi.cpu().numpy()
    np.savez(sys.argv[5], **outs); sys.exit(0)

bad = 0
cases = adi_cases()
with tempfile.TemporaryDirectory() as d:
    json.dump(cases, open(os.path.join(d, 'cases.json'), 'w'))
    res = []
    for flag in ('1', '0'):
        o = os.path.join(d, 'o%s.npz' % flag)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(seed), str(N), '--adi-child', os.path.join(d, 'cases.json'), o], env=dict(os.environ, NNS_ADI_LDS=flag),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(o))
    for k, c in enumerate(cases):
        same = all(np.array_equal(res[0][n + str(k)], res[1][n + str(k)]) and np.isfinite(res[0][n + str(k)]).all() for n in 'uv')
        bad += not same
        print('adi   %s %s' % (c, 'ok' if same else 'BAD'), flush=True)

import torch
from nns import ops
from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as Nm
from nns.chorin_fd import NavierStokesSystem
sides = ['left', 'right', 'top', 'bottom']
def rand_bcs(dx, dy):
    return [(D if rng.integers(0, 2) else Nm)(float(rng.uniform(-1, 1)), str(rng.choice(sides)), dx, dy) for _ in range(int(rng.integers(1, 7)))]
for _ in range(N):
    n = int(rng.integers(5, 65)); B = int(rng.integers(1, 4)); f64 = bool(rng.integers(0, 2)); nit = int(rng.choice([2, 5, 20, 50])); adv = str(rng.choice(['reference', 'corrected']))
    dx = dy = 2. / (n - 1)
    bcs = [rand_bcs(dx, dy) for _ in range(3)]
    shape = (n, n) if B == 1 else (B, n, n)
    ic = [0.05 * rng.standard_normal(shape) for _ in range(3)]
    runs = []
    for fused in (True, False):
        s = NavierStokesSystem(ic[0].copy(), ic[1].copy(), ic[2].copy(), *bcs, nt=6, nit=nit, nx=n, ny=n, dt=1e-3, rho=1.1, nu=0.05, beta=1.25, method='explicit',
                               dtype=np.float64 if f64 else np.float32, advection=adv)
        s.fused_step = fused
        runs.append(list(s.simulate_device()) + [s.last_sor_info.clone()])
    same = all(torch.equal(a, b) for a, b in zip(*runs))
    bad += not same
    print('step  n %d B %d f64 %d nit %d %s bcs %s %s' % (n, B, f64, nit, adv, [len(b) for b in bcs], 'ok' if same else 'BAD'), flush=True)
for _ in range(N):
    nx, ny = int(rng.integers(3, 80)), int(rng.integers(3, 80)); B = int(rng.integers(1, 4)); dt_ = torch.float64 if rng.integers(0, 2) else torch.float32
    nit, tol = int(rng.integers(1, 150)), float(rng.choice([1e-3, 0.05, 0.5]))
    p0 = torch.as_tensor(rng.standard_normal((B, nx, ny)), dtype=dt_, device='cuda'); C = torch.as_tensor(rng.standard_normal((B, nx, ny)) * 5, dtype=dt_, device='cuda')
    ref = p0.clone(); iref = ops.fd_sor_(ref, C, 0.03, 0.04, 1.25, tol, nit)
    hint = torch.as_tensor(np.stack([rng.integers(-3, 200, size=B), np.zeros(B)], 1), dtype=dt_, device='cuda')
    p = p0.clone(); info = ops.fd_sor_(p, C, 0.03, 0.04, 1.25, tol, nit, hint=hint)
    same = torch.equal(p, ref) and torch.equal(info, iref)
    bad += not same
    print('hint  %d x %d B %d %s nit %d tol %g sweeps %s %s' % (nx, ny, B, str(dt_)[6:], nit, tol, iref[:, 0].tolist(), 'ok' if same else 'BAD'), flush=True)
print('%d bad' % bad)
sys.exit(1 if bad else 0)
