"""Timing of nns_coarsen_* (u, v, p in one launch): achieved HBM GB/s on [64, 1024, 1024] sequences."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import torch
from nns import ops
out = {}
for dt in (torch.float32, torch.float64):
    f = [torch.randn(64, 1024, 1024, device='cuda', dtype=dt) for _ in range(3)]
    for ax, ay in ((4, 4), (2, 2), (8, 8), (16, 16)):
        for _ in range(3): ops.coarsen(*f, ax, ay)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ops.coarsen(*f, ax, ay)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
        b = 3 * f[0].numel() * f[0].element_size() * (1 + 1.0 / (ax * ay))
        out['%s_%dx%d' % (str(dt).split('.')[1], ax, ay)] = dict(ms=1e3 * t, GBs=b / t / 1e9)
print(json.dumps(out))
