"""Developer helper: the fused spectral residual backward at 1024^2 x 64 (for rocprofv3 / A-B timing)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from nns import ops
n, B = 1024, 64
f = [torch.randn(B, n, n, device='cuda') for _ in range(5)]
L = 2 * np.pi
out = {}
PROFILE = os.environ.get('NNS_PROFILE', '0') == '1'            # under rocprofv3 (tools/prof_any.sh): the library's pick only, few launches
cases = (('library pick (all-float32 here)', 1),) if PROFILE else (('library pick (all-float32 here)', 1), ('float64 forward', 2), ('library pick (all-float32 here) ', 1), ('float64 forward ', 2))
for name, prec in cases:
    for _ in range(1 if PROFILE else 3):
        ops.spec_residual_bwd(*f, 1e-3, L, L, 1.0, L / 1000, precise=prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4 if PROFILE else 20):
        ops.spec_residual_bwd(*f, 1e-3, L, L, 1.0, L / 1000, precise=prec)
    torch.cuda.synchronize()
    out[name] = 1e3 * (time.perf_counter() - t0) / (4 if PROFILE else 20)
print(json.dumps(dict(spec_bwd_ms=out)))
