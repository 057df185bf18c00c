"""Developer probe: where does the host stall in the cfg 5 step?  Host-side durations of the individual calls (no device sync in between)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import torch
from nns import ops
import bench_configs
bench_configs.secondary(cpu=False)
K, n, nt, mb = 10, 256, 32, 256
coeff = torch.randn(nt * mb, K, 3, device='cuda'); basis = torch.randn(K, 3, n * n, device='cuda')
obs = torch.randn(nt * mb, 3, n * n, device='cuda')
torch.cuda.synchronize()
worst = []
for i in range(60):
    t0 = time.perf_counter()
    r = ops.basis_loss_fused(coeff, basis, obs)
    t1 = time.perf_counter()
    y = torch.sqrt(r[0]); z = y * 2.0; w = z + 1.0
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    worst.append((round(1e3 * (t1 - t0), 2), round(1e3 * (t2 - t1), 2), round(1e3 * (t3 - t2), 2)))
print('host ms per iteration (our call, three torch ops, sync):')
print([w for w in worst if max(w) > 5.0], 'of', len(worst), '; typical', worst[-1])
