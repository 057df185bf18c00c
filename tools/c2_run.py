"""Developer helper: the BASELINE config 2 training iteration (for `rocprofv3 --kernel-trace --stats`): which launches make up its ~1.1 ms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import torch
from nns.neural_spectral.spectral_ode import PDEFunc
import nns.optim as nns_optim
K, n, nt = 10, 128, 100
m = PDEFunc(K, n, n).cuda()
obs = torch.randn(nt, 1, 3, n, n, device='cuda')
t = torch.arange(nt, device='cuda') + 1
opt = nns_optim.Adam(m.parameters(), lr=1e-3)
def it():
    opt.zero_grad()
    m.loss(obs[0], t, obs).backward()
    opt.step()
for _ in range(3): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): it()
torch.cuda.synchronize(); print('cfg2 training iteration: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
