"""Developer helper: the BASELINE config 5 training step (ensemble 256 x 256^2, K = 10, nt = 32) -- PDEFunc.loss(...).backward() through
the one-sweep fused basis loss -- for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neural-navier-stokes_amd'))
import torch
from nns.neural_spectral.spectral_ode import PDEFunc
K, n, nt, mb = 10, 256, 32, 256
m = PDEFunc(K, n, n).cuda()
obs = torch.randn(nt, mb, 3, n, n, device='cuda')
t = torch.arange(nt, device='cuda') + 1
for _ in range(4):
    m.zero_grad()
    m.loss(obs[0], t, obs).backward()
torch.cuda.synchronize()
