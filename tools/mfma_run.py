"""Developer helper: run ONE family of matrix-core kernels a few times, for the rocprofv3 counter passes of tools/mfma_pmc.sh.
    python3 tools/mfma_run.py pm    depth-8 width-64 pixel MLP at 16 x 512^2: forward bf16, forward float32, fused backward (BASELINE config 3)
    python3 tools/mfma_run.py c5    ensemble step 256 x 256^2, K = 10, nt = 32: the one-sweep loss + gradient kernel on exact float32 MFMA (config 5)
    python3 tools/mfma_run.py ode   neural_spectral training iteration 128^2, K = 10, nt = 100 (config 2) + the ODE MLP at mb = 8192 (tile kernel)
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neural-navier-stokes_amd'))
import torch
from nns import ops
which = sys.argv[1]
if which == 'pm':
    from nns.neural_spectral.spectral_ode import PixelMLP
    m = PixelMLP(8, 64).cuda()
    x = torch.randn(16, 3, 512, 512, device='cuda')
    gy = torch.randn_like(x)
    ws, bs = [w.detach() for w in m.weights], [b.detach() for b in m.biases]
    for _ in range(6):
        m(x, bf16=True); m(x, bf16=False); ops.pixel_mlp_bwd(x, gy, ws, bs)
elif which == 'c5':
    from nns.neural_spectral.spectral_ode import PDEFunc
    K, n, nt, mb = 10, 256, 32, 256
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, mb, 3, n, n, device='cuda')
    t = torch.arange(nt, device='cuda') + 1
    for _ in range(5):
        m.zero_grad()
        m.loss(obs[0], t, obs).backward()
elif which == 'ode':
    from nns.neural_spectral.spectral_ode import PDEFunc, ODEFunc
    from nns.neural_spectral.anode import odesolver_adjoint
    K, n, nt = 10, 128, 100
    m = PDEFunc(K, n, n).cuda()
    obs = torch.randn(nt, 1, 3, n, n, device='cuda')
    t = torch.arange(nt, device='cuda') + 1
    for _ in range(5):
        m.zero_grad()
        m.loss(obs[0], t, obs).backward()
    f = ODEFunc(K).cuda()
    z0 = torch.randn(8192, 3 * K, device='cuda', requires_grad=True)          # a batch large enough for the 16-row MFMA tile kernels
    for _ in range(3):
        out = odesolver_adjoint(f, z0, options=dict(Nt=20, method='RK4'))
        out.sum().backward()
torch.cuda.synchronize()
