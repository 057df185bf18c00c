"""Developer helper (GPU box): a soak run of the hot paths -- many back-to-back evaluations with the outputs compared BITWISE against the first
one (the kernels are deterministic: no atomics on the residual paths), device memory watched for growth, step-time percentiles (the pool's boxes
show sporadic 40-70 ms device stalls: bench_configs.timeit3), then a long physics-informed training run and a long cavity run (finite, monotone where
it should be)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns import ops
from nns.periodic import ResidualEngine
from nns.synthetic import residual_inputs
out = {}
n, B = 1024, 64
f = [torch.as_tensor(np.tile(a, (B // 4, 1, 1)), device='cuda') for a in residual_inputs(4, n)]
eng = ResidualEngine(n, n, 1e-3, 1.0, 2 * np.pi / 1000)
o_fd = tuple(torch.empty_like(f[0]) for _ in range(3)); o_sp = tuple(torch.empty_like(f[0]) for _ in range(3))
eng.both(*f, out_fd=o_fd, out_spec=o_sp); torch.cuda.synchronize()
ref = [t.clone() for t in o_fd + o_sp]
mem0 = torch.cuda.memory_allocated()
NBLK, PER = int(os.environ.get('SOAK_BLOCKS', '200')), 100
times, mismatches = [], 0
for blk in range(NBLK):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(PER):
        eng.both(*f, out_fd=o_fd, out_spec=o_sp)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / PER * 1e3)
    if blk % 10 == 9:
        mismatches += sum(not torch.equal(a, b) for a, b in zip(o_fd + o_sp, ref))
times = np.array(times)
out['headline'] = dict(evaluations=NBLK * PER, ms_per_step=dict(min=float(times.min()), median=float(np.median(times)), p99=float(np.percentile(times, 99)), max=float(times.max())),
                       blocks_slower_than_1p5x_median=int((times > 1.5 * np.median(times)).sum()), bitwise_mismatches=int(mismatches),
                       device_memory_growth_bytes=int(torch.cuda.memory_allocated() - mem0))
print(json.dumps(out['headline']), flush=True)
# backward of the spectral residual: same checks
g = [torch.randn_like(f[0]) for _ in range(3)]
r0 = [t.clone() for t in ops.spec_residual_bwd(f[0], f[1], *g, 1e-3, 2 * np.pi, 2 * np.pi, 1.0, 2 * np.pi / 1000)[:3]]
mm = 0
for it in range(300):
    r = ops.spec_residual_bwd(f[0], f[1], *g, 1e-3, 2 * np.pi, 2 * np.pi, 1.0, 2 * np.pi / 1000)
    if it % 50 == 49: mm += sum(not torch.equal(a, b) for a, b in zip(r[:3], r0))
out['spectral_backward'] = dict(evaluations=300, bitwise_mismatches=int(mm))
print(json.dumps(out['spectral_backward']), flush=True)
del f, o_fd, o_sp, ref, g, r0, r
torch.cuda.empty_cache()
# physics-informed training, 1500 steps at 8 x 256^2
from nns.neural_spectral.physics_informed import FieldStepper, train_step
import nns.optim as nns_optim
m = 256
ri = residual_inputs(8, m)
state = torch.as_tensor(np.stack(ri[3:] + ri[2:3], axis=1), device='cuda'); target = torch.as_tensor(np.stack(ri[:3], axis=1), device='cuda')
torch.manual_seed(0)
stepper = FieldStepper(8, 64).cuda(); opt = nns_optim.Adam(stepper.parameters(), lr=3e-4)
e2 = ResidualEngine(m, m, 1e-3, 1.0, 2 * np.pi / 1000, backend='fd9')
mem1 = torch.cuda.memory_allocated(); hist = []
for it in range(1500):
    l = train_step(stepper, e2, opt, state, target, lam=0.1)
    if it % 100 == 0 or it == 1499: hist.append(float(l[0]))
out['physics_informed_training'] = dict(steps=1500, loss_every_100=hist, finite=bool(np.isfinite(hist).all()), decreased=bool(hist[-1] < 0.5 * hist[0]),
                                        device_memory_growth_bytes=int(torch.cuda.memory_allocated() - mem1))
print(json.dumps(out['physics_informed_training']), flush=True)
# cavity, 3000 steps: steady state approached (the update between consecutive steps shrinks), finite
from nns.chorin_fd import NavierStokesSystem
from nns.boundary import DirichletBoundaryCondition as D, NeumannBoundaryCondition as N
c = 64; dx = dy = 2. / (c - 1)
u_bc = [D(0, 'left', dx, dy), D(1, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
v_bc = [D(0, 'left', dx, dy), D(0, 'right', dx, dy), D(0, 'top', dx, dy), D(0, 'bottom', dx, dy)]
p_bc = [D(0, 'top', dx, dy), N(0, 'bottom', dx, dy), N(0, 'left', dx, dy), N(0, 'right', dx, dy)]
z = np.zeros((c, c))
s = NavierStokesSystem(z, z.copy(), z.copy(), u_bc, v_bc, p_bc, nt=3000, nit=50, nx=c, ny=c, dt=1e-3, rho=1, nu=0.02, beta=1.25, method='explicit')
t0 = time.perf_counter(); us, vs, ps = s.simulate_device(); torch.cuda.synchronize(); dtc = time.perf_counter() - t0
d = [(us[k] - us[k - 1]).abs().max().item() for k in (10, 1000, 2999)]
out['cavity_64_3000_steps'] = dict(seconds=dtc, ms_per_step=dtc / 3000 * 1e3, finite=bool(torch.isfinite(us[-1]).all() and torch.isfinite(ps[-1]).all()), max_update_at_steps_10_1000_2999=d,
                                   sor_sweeps_last_step=s.sor_info()[0][0])
print(json.dumps(out['cavity_64_3000_steps']), flush=True)
print('SOAK ' + json.dumps(out))
