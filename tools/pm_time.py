"""Developer helper: time the depth-8 width-64 bf16 pixel-MLP forward and backward at 16 x 512^2 pixels."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neural-navier-stokes_amd'))
import torch
from nns import ops
from nns.neural_spectral.spectral_ode import PixelMLP
m = PixelMLP(8, 64).cuda()
x = torch.randn(16, 3, 512, 512, device='cuda'); gy = torch.randn_like(x)
ws, bs = [w.detach() for w in m.weights], [b.detach() for b in m.biases]
def tm(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / it
print(json.dumps(dict(fwd_bf16_ms=tm(lambda: m(x, bf16=True)), bwd_bf16_ms=tm(lambda: ops.pixel_mlp_bwd(x, gy, ws, bs)))))
