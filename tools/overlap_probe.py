"""Developer probe (round 3): can the column pass (vector-issue / L1 bound, HBM mostly idle) of batch piece q+1 run UNDER the fused row pass
(HBM-bound) of piece q?  Two streams, Q pieces of the batch:  main: x(0) r(0) r(1) ... ;  side: x(1) x(2) ... with events so that x(q+1) starts
when x(q) is done and r(q) waits for x(q).  Prints ms per step for the plain two-launch step and for Q in {2, 4, 8}."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neural-navier-stokes_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from nns import ops
from bench import make_inputs
n, B = 1024, 64
dt, nu, rho, L = 1e-3, 2 * np.pi / 1000, 1.0, 2 * np.pi
dev = torch.device('cuda', 0)
f = make_inputs(B, n, 8, 1234, dev)
ofd = tuple(torch.empty_like(f[0]) for _ in range(3)); osp = tuple(torch.empty_like(f[0]) for _ in range(3))
ref_fd, ref_sp = ops.residual_both(*f, dt, L, L, rho, nu, 1)
side = torch.cuda.Stream()


def plain():
    ops.residual_both(*f, dt, L, L, rho, nu, 1, out_fd=ofd, out_spec=osp)


def piped(Q):
    main = torch.cuda.current_stream()
    bounds = [(B * q // Q, B * (q + 1) // Q) for q in range(Q)]
    sl = [slice(*b) for b in bounds]
    start = torch.cuda.Event(); start.record(main)
    xdone = [torch.cuda.Event() for _ in range(Q)]
    # column pass of piece 0 on the main stream, the others on the side stream, each after the previous one
    ops.spec_residual_xpass(f[0][sl[0]], f[1][sl[0]], f[2][sl[0]], L, rho, nu, 1, out=tuple(t[sl[0]] for t in osp))
    xdone[0].record(main)
    with torch.cuda.stream(side):
        side.wait_event(start)
        for q in range(1, Q):
            side.wait_event(xdone[q - 1])
            ops.spec_residual_xpass(f[0][sl[q]], f[1][sl[q]], f[2][sl[q]], L, rho, nu, 1, out=tuple(t[sl[q]] for t in osp))
            xdone[q].record(side)
    for q in range(Q):
        main.wait_event(xdone[q])
        ops.residual_both(*[t[sl[q]] for t in f], dt, L, L, rho, nu, 1, out_fd=tuple(t[sl[q]] for t in ofd), out_spec=tuple(t[sl[q]] for t in osp), rowpass_only=True)


def timeit(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters


print('plain two-launch step: %.4f ms' % timeit(plain))
for Q in (2, 4, 8):
    t = timeit(lambda: piped(Q))
    ok = all(torch.equal(a, b) for a, b in zip(list(ofd) + list(osp), list(ref_fd) + list(ref_sp)))
    print('Q = %d pieces, column pass of q+1 under the row pass of q: %.4f ms  (bitwise equal to the plain step: %s)' % (Q, t, ok))
print('plain again: %.4f ms' % timeit(plain))
