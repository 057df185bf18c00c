"""Developer helper: run the depth-8 width-64 pixel MLP (forward bf16, forward fp32, backward) a few times, for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neural-navier-stokes_amd'))
import torch
from nns import ops
from nns.neural_spectral.spectral_ode import PixelMLP
m = PixelMLP(8, 64).cuda()
x = torch.randn(16, 3, 512, 512, device='cuda')
gy = torch.randn_like(x)
ws, bs = [w.detach() for w in m.weights], [b.detach() for b in m.biases]
for _ in range(4):
    m(x, bf16=True); m(x, bf16=False); ops.pixel_mlp_bwd(x, gy, ws, bs)
torch.cuda.synchronize()
