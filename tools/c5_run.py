"""Developer helper: the fused basis loss forward / backward at the BASELINE config 5 shape, for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neural-navier-stokes_amd'))
import torch
from nns import ops
K, n, nt, mb = 10, 256, 32, 256
T, C, P = nt * mb, 3, n * n
coeff = torch.randn(T, K, C, device='cuda'); basis = torch.randn(K, C, P, device='cuda'); obs = torch.randn(T, C, P, device='cuda')
for _ in range(3):
    ops.basis_loss_fwd(coeff, basis, obs); ops.basis_loss_bwd(coeff, basis, obs, 0.5)
torch.cuda.synchronize()
