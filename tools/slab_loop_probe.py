"""Developer helper (one GPU): the slab step on the world-1 RCCL loopback, timed in alternating blocks of chunk counts -- where does a
step's wall time go when the collectives are device-to-device copies?  `python tools/slab_loop_probe.py [blocks]`"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from nns.slab import SlabResidual
n, B = 1024, 64
L = 2 * np.pi
f = [torch.randn(B, n, n, device='cuda') for _ in range(5)]
sl = SlabResidual(n, n, 1e-3, 1.0, L / 1000, L, L, precise=1, loopback=True)
FIXED = os.environ.get('NNS_PROBE_FIXED', '1') == '1'      # caller-owned outputs: recorded C calls are replayed
ofd = tuple(torch.empty_like(f[0]) for _ in range(3)); osp = tuple(torch.empty_like(f[0]) for _ in range(3))
kw = dict(out_fd=ofd, out_spec=osp) if FIXED else {}
PROFILE = os.environ.get('NNS_PROFILE', '0') == '1'
out = []
for rnd in range(1 if PROFILE else int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for chunks in (1, 2, 4):
        for _ in range(2 if PROFILE else 5):
            sl.both(*f, chunks=chunks, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 4 if PROFILE else 20
        for _ in range(K):
            sl.both(*f, chunks=chunks, **kw)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        out.append(dict(chunks=chunks, ms_per_step=1e3 * (time.perf_counter() - t0) / K, host_ms=1e3 * th / K))
# what bench.py does around its timed region: a device synchronisation, a barrier on the process group, a synchronisation
def bench_like(chunks, K=20, barrier=True):
    for _ in range(5):
        sl.both(*f, chunks=chunks, **kw)
    torch.cuda.synchronize()
    if barrier:
        dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        sl.both(*f, chunks=chunks, **kw)
    torch.cuda.synchronize()
    if barrier:
        dist.barrier(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K
if not PROFILE:
    for chunks in (1, 2, 1):
        out.append(dict(bench_like=True, barrier=True, chunks=chunks, ms_per_step=bench_like(chunks)))
        out.append(dict(bench_like=True, barrier=False, chunks=chunks, ms_per_step=bench_like(chunks, barrier=False)))
print(json.dumps(out))
dist.destroy_process_group()
