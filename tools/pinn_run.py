"""Developer helper: the physics-informed training step at the BASELINE config 3 shape (for rocprofv3 / timing)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns.periodic import ResidualEngine
import nns.optim as nns_optim
from nns.synthetic import residual_inputs
from nns.neural_spectral.physics_informed import FieldStepper, train_step
n = 512
layout = sys.argv[1] if len(sys.argv) > 1 else 'bchw'
backend = sys.argv[2] if len(sys.argv) > 2 else 'fd9'          # 'fd9' | 'fd5' | 'spectral'
ri = residual_inputs(4, n)
state = torch.as_tensor(np.stack([np.tile(a, (4, 1, 1)) for a in ri[3:] + ri[2:3]], axis=1), device='cuda')
target = torch.as_tensor(np.stack([np.tile(a, (4, 1, 1)) for a in ri[:3]], axis=1), device='cuda')
if layout == 'cm':
    state, target = state.transpose(0, 1).contiguous(), target.transpose(0, 1).contiguous()
stepper = FieldStepper(8, 64).cuda()
opt = nns_optim.Adam(stepper.parameters(), lr=1e-4)
eng = ResidualEngine(n, n, 1e-3, 1.0, 2 * np.pi / 1000, backend=backend)
fused = os.environ.get('NNS_PINN_FUSED', '1') != '0'
for _ in range(3): out = train_step(stepper, eng, opt, state, target, lam=0.1, layout=layout, fused=fused)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): out = train_step(stepper, eng, opt, state, target, lam=0.1, layout=layout, fused=fused)
torch.cuda.synchronize()
print(json.dumps(dict(layout=layout, backend=backend, fused_head=fused, step_ms=1e3 * (time.perf_counter() - t0) / 10, loss=[float(x) for x in out])))
