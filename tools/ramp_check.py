"""Developer helper (GPU box): the headline step timed in consecutive blocks of 20 right after start-up -- does the rate settle?
(Round 2: block 0 after 5 warm-up steps 1.311 ms/step, every later block 1.268 +- 0.003: bench.py defaults to 30 warm-up steps.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np, torch
sys.argv = ['bench.py']
import bench
from nns.periodic import ResidualEngine
n, B = 1024, 64
f = bench.make_inputs(B, n, 8, 1234, torch.device('cuda', 0))
eng = ResidualEngine(n, n, 1e-3, 1.0, 2 * np.pi / 1000, 2 * np.pi, 2 * np.pi, backend='spectral', precise=1)
out_fd = tuple(torch.empty_like(f[0]) for _ in range(3)); out_sp = tuple(torch.empty_like(f[0]) for _ in range(3))
def step(): eng.both(*f, out_fd=out_fd, out_spec=out_sp, stencil=5)
for _ in range(5): step()
torch.cuda.synchronize()
t_start = time.perf_counter()
for blk in range(40):
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('block %2d at %.2f s: %.3f ms/step' % (blk, t0 - t_start, (t1 - t0) / 20 * 1e3))
