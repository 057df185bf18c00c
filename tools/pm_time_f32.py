import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'neural-navier-stokes_amd'))
import torch
from nns.neural_spectral.spectral_ode import PixelMLP
m = PixelMLP(8, 64).cuda()
x = torch.randn(16, 3, 512, 512, device='cuda')
for _ in range(3): m(x, bf16=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m(x, bf16=False)
torch.cuda.synchronize(); print('forward f32: %.3f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
