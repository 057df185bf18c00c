"""Developer probe: per-iteration wall times of the cfg 5 step after the other secondary configs ran in the same process (bench.py's order)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import torch
import bench_configs
o = bench_configs.secondary(cpu=False)
print('secondary cfg5 ms', o['cfg5_ensemble_256x256_step']['ms'])
from nns.neural_spectral.spectral_ode import PDEFunc
K, n, nt, mb = 10, 256, 32, 256
m = PDEFunc(K, n, n).cuda()
obs = torch.randn(nt, mb, 3, n, n, device='cuda')
t = torch.arange(nt, device='cuda') + 1
ts = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.zero_grad(); m.loss(obs[0], t, obs).backward()
    torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 3))
print('per-iteration ms', ts)
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
