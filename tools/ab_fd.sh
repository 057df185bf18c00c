#!/bin/bash
# Developer helper (GPU box): same-box A/B of the standalone FD residual kernel over library variants (ab_variants/libnns_hip_<tag>.so)
cd $GRAFT_REPO_ROOT
cat > /tmp/fdt.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'neural-navier-stokes_amd'))
import numpy as np, torch
from nns import ops
n, B = 1024, 64
f = [torch.randn(B, n, n, device='cuda') for _ in range(5)]
out = tuple(torch.empty_like(f[0]) for _ in range(3))
h = 2 * np.pi / n
def tm(st):
    for _ in range(5): ops.fd_residual(*f, 1e-3, h, h, 1.0, 0.006, st, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ops.fd_residual(*f, 1e-3, h, h, 1.0, 0.006, st, out=out)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / 30
print(json.dumps(dict(fd5_ms=round(tm(5), 4), fd9_ms=round(tm(9), 4))))
PY
for r in 1 2; do for t in main "$@"; do
  if [ $t = main ]; then unset NNS_LIB_PATH; else export NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_$t.so; fi
  echo -n "$t "; python /tmp/fdt.py 2>/dev/null
done; done | tee gpurun_out/ab_fd.log
