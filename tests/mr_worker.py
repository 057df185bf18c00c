"""One rank of the multi-rank GPU tests (tests/test_gpu_multirank.py): started as a FRESH child process (RANK, WORLD_SIZE,
MASTER_ADDR, MASTER_PORT in the environment), transport gloo with EVERY rank on cuda:0 (a one-GPU box) by default; MR_BACKEND=nccl MR_DEVICE=rank = RCCL, one rank per GPU, compute = the HIP
back-ends of nns.slab / nns.data_parallel.  Writes this rank's results to <out>/<case>_r<rank>.npz.

    python tests/mr_worker.py <case> <out-dir>        case in {residual, chorin, ensemble}
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

import mr_cases as MC


def main():
    case, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    backend = os.environ.get('MR_BACKEND', 'gloo')              # 'nccl' (= RCCL): one rank per GPU, or the world-1 loopback case
    dev = rank if os.environ.get('MR_DEVICE', '0') == 'rank' else 0
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        res = getattr(MC, 'rank_' + case)(rank, world)
        np.savez(os.path.join(out, '%s_r%d.npz' % (case, rank)), **res)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
