#!/usr/bin/env python3
"""Headline benchmark: grid-point residual-updates/sec at 1024^2 (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
the incompressible Navier-Stokes residual (r_u, r_v, r_div) of B = 64 independent 1024x1024
periodic boxes evaluated with BOTH derivative back-ends on the same inputs -- the 5-point finite-
difference stencil (csrc/residual_kernels.hip) and the Fourier-spectral path (two LDS-resident FFT
passes, csrc/spectral_kernels.hip).  One residual update = one grid point through both.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: N fresh child
processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU) before anything in this process has touched the
GPU, relays rank 0's JSON line and exits with the children's status.  Under `torch.distributed.run` (WORLD_SIZE set) it
is one of the ranks.

N > 1, default `--mode slab` (BASELINE.json's north star, config 4): the SAME 64 grids are slab-decomposed by rows over
the ranks -- nearest-neighbour halo send/recv for the stencil, two all-to-all transposes per spectral evaluation, batch
chunks pipelined so the collectives overlap the kernels (nns/slab.py) -- "scaling": "strong", `value` = 64 x 1024^2
points / step time (max over ranks).  The same run also reports, as `batch_sharded`, the embarrassingly parallel
alternative (every rank its own 64 grids, no data-path collective, weak scaling), `transport` (the backend really used),
`rccl_ranks` (a device all-reduce of ones) and `phases` (per-phase times of one un-pipelined evaluation).  A slab run that
fails or hangs makes the launcher exit non-zero with the ranks' logs on stderr (the cause is to be found and fixed, not papered over);
only with `--fallback-batch` does it re-run the ranks in `--mode batch`, and then the line says so in `launcher` and carries the
batch number under `value` with `scaling: weak`.  Both attempts and the teardown fit 540 s (`--launch-timeout`, default 240 s each).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- for the dominant kernel: algorithmic bytes per launch / its average launch time,
                  measured here with HIP events on the launch stream, against the 8 TB/s HBM peak;
  cpu_baseline -- the NumPy oracle (oracle/periodic.py, kind "port") timed on this host on a bounded
                  sample of the same workload: one core (the headline row, "cores": 1) and, as
                  cpu_baseline.all_cores, one oracle process per usable host core (at most 16).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable
# algorithmic (compulsory) HBM bytes per grid point, float32 fields -- derivation in DESIGN.md
BYTES_PER_PT = {'fd_residual': 32.0, 'spec_xpass': 24.0, 'spec_ypass': 44.0, 'both_rowpass_back_to_back': 56.0,
                # fused row pass (nns_residual_both_f32): u, v, p, u_prev, v_prev + 3 column-pass partials in, 3 + 3 residuals out
                'both_rowpass': 56.0}
STEP_BYTES_TWO_PASS = 80.0     # column pass (3 in + 3 partials out) + fused row pass (5 in + 3 partials in + 6 out)
STEP_BYTES_COMPULSORY = 44.0   # u, v, p, u_prev, v_prev in; 3 FD + 3 spectral residual fields out


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(batch, n, distinct, seed0, device):
    from nns.synthetic import residual_inputs
    d = max(1, min(distinct, batch))
    f = residual_inputs(d, n, seed0=seed0)
    reps = (batch + d - 1) // d
    return [torch.as_tensor(np.tile(a, (reps, 1, 1))[:batch], device=device).contiguous() for a in f]


def time_kernel(fn, iters):
    """Average launch duration (ms) of fn() over `iters` back-to-back launches, HIP events on the
    current stream (the stream the kernels are enqueued on)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_baseline(n, budget_s=20.0, check=None):
    """The NumPy oracle (float64) on this host: FD 5-point + spectral residual of single 1024^2 grids,
    repeated until ~budget_s of CPU time.  check: the timed step's six output fields of grid 0 (the same inputs): their rel-L2
    distance from the oracle is reported as cpu_baseline.oracle_check (the oracle as the checker of the measured path)."""
    from oracle import periodic as OP
    from nns.synthetic import residual_inputs
    f = [a[0].astype(np.float64) for a in residual_inputs(1, n)]
    dt, nu, rho, L = 1e-3, 2 * np.pi / 1000, 1.0, 2 * np.pi
    h = L / n
    oracle_check = None
    if check is not None:
        ref = list(OP.fd_residual(*f, dt, h, h, rho, nu, 5)) + list(OP.spectral_residual(*f, dt, L, L, rho, nu))
        rl = [float(np.linalg.norm(g - r) / np.linalg.norm(r)) for g, r in zip(check, ref)]
        oracle_check = dict(fd_rel_l2=rl[:3], spectral_rel_l2=rl[3:], tolerance=1e-5, grid='grid 0 of the timed batch, outputs of the last timed step')
    t0 = time.perf_counter()
    reps = 0
    while True:
        OP.fd_residual(*f, dt, h, h, rho, nu, 5)
        OP.spectral_residual(*f, dt, L, L, rho, nu)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 400:
            break
    return dict(value=reps * n * n / el, unit='residual-updates/s', cores=1, kind='port',
                sample='%d x (FD 5-point + rfft2 spectral residual) of one %dx%d float64 grid, NumPy oracle, %.1f s'
                       % (reps, n, n, el), host_cores_present=os.cpu_count(), oracle_check=oracle_check)


def _cpu_worker(arg):
    """One host core: the same oracle loop as cpu_baseline on its own grid, for ~budget_s seconds."""
    n, seed, budget_s = arg
    from oracle import periodic as OP
    from nns.synthetic import residual_inputs
    f = [a[0].astype(np.float64) for a in residual_inputs(1, n, seed0=1234 + seed)]
    dt, nu, rho, L = 1e-3, 2 * np.pi / 1000, 1.0, 2 * np.pi
    h = L / n
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s:
        OP.fd_residual(*f, dt, h, h, rho, nu, 5)
        OP.spectral_residual(*f, dt, L, L, rho, nu)
        reps += 1
    return reps, time.perf_counter() - t0


def cpu_baseline_all_cores(n, budget_s=8.0, max_procs=16):
    """The oracle on all the host cores this job may use (one process per core, one grid each: element-wise NumPy and
    pocketfft are single-threaded).  Runs BEFORE anything touches the GPU: the workers are forked."""
    import multiprocessing as mp
    procs = max(1, min(max_procs, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))
    with mp.get_context('fork').Pool(procs) as pool:
        res = pool.map_async(_cpu_worker, [(n, i, budget_s) for i in range(procs)]).get(timeout=6 * budget_s + 30)   # never hang the bench line
    rate = sum(r * n * n / t for r, t in res)
    return dict(value=rate, unit='residual-updates/s', cores=procs,
                sample='%d processes x (FD 5-point + rfft2 spectral residual) of one %dx%d float64 grid each, %.0f s' % (procs, n, n, budget_s))


def slab_phases(sl, f, iters):
    """One un-pipelined SlabResidual.both evaluation, phase by phase (ms, mean over `iters`): HIP events on the launch stream around
    every phase, the collectives waited on that stream, a device synchronisation between phases.  Same kernels and messages as
    the timed step, without the overlap."""
    u, v, p, up, vp = f
    B, nloc, ny = u.shape
    P, nyl, c = sl.P, sl.nyloc, sl.compute
    shape = (P, 3, B, nloc, nyl)
    send, recv, back, got = (sl._buf(('ph', i), shape, u) for i in range(4))
    first, last, top, bot = sl._halo_bufs(u, 3, 'ph')
    out_fd = tuple(torch.empty_like(u) for _ in range(3))
    out_sp = tuple(torch.empty_like(u) for _ in range(3))
    acc = {}

    def timed(name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record()
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
        return r
    for it in range(iters + 1):
        if it == 1:
            acc.clear()                                        # first round = warm-up
        timed('pack_with_halo_rows', lambda: c.pack_halo([u, v, p], send, first, last, 0, P))
        h = timed('halo_post', lambda: sl.tr.ring_exchange(first, last, bot, top, wrap=True))
        timed('all_to_all_1', lambda: sl.tr.all_to_all(recv, send).wait())
        timed('column_pass', lambda: c.spec_xpass_seg(recv, back, B, sl.nx, nyl, nloc, sl.Lx, sl.rho, sl.nu, sl.precise))
        timed('all_to_all_2', lambda: sl.tr.all_to_all(got, back).wait())
        timed('halo_wait', lambda: h.wait())
        timed('row_pass_on_receive_buffer', lambda: c.both_rowpass_halo_seg(u, v, p, up, vp, top, bot, got, sl.dt, sl.dx, sl.Ly, sl.rho, sl.nu, sl.precise, out_fd, out_sp))
    return {k: v / iters for k, v in acc.items()}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _spawn_ranks(n, argv, timeout_s, script=None):
    """Start n fresh child ranks of this script (one per GPU), wait for them (bounded), return (rc, rank 0's stdout lines, note).
    Children are ended by their exact PIDs when one fails or the time is up."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=(r == 0)))
    import threading
    lines = []
    rd = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().splitlines()), daemon=True)
    rd.start()
    t_end = time.monotonic() + timeout_s
    note, rc = None, 0
    while True:
        codes = [q.poll() for q in procs]
        if all(c is not None for c in codes):
            rc = next((c for c in codes if c != 0), 0)
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.monotonic() > t_end:
            note = ('rank %d exited with status %d' % bad[0]) if bad else ('no result after %.0f s' % timeout_s)
            time.sleep(2.0 if bad else 0.0)                           # let the other ranks notice and report
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            t_kill = time.monotonic() + 10
            while any(q.poll() is None for q in procs) and time.monotonic() < t_kill:
                time.sleep(0.2)
            for q in procs:
                if q.poll() is None:
                    q.kill()
            rc = bad[0][1] if bad else 124
            break
        time.sleep(0.2)
    rd.join(timeout=5)
    return rc, lines, note


LAUNCH_TOTAL_S = 540.0           # everything the self-launcher does (all attempts + teardown) fits the driver's 600 s bench limit


def self_launch(args, argv=None, script=None):
    """`python bench.py --gpus N` (N > 1) outside a launcher: run the N ranks as child processes and relay rank 0's JSON line.
    ONE attempt in the requested mode (default slab).  If it fails or hangs, the launcher exits NON-ZERO with the failure on stderr: a hang in
    the multi-GPU path must surface as a failure so that its cause is found from the logs and fixed.  With `--fallback-batch` a failed
    slab attempt is followed by a batch-mode attempt in fresh processes; the line then says so (`launcher.failed_attempts`, `scaling: weak`).
    Time: every attempt is cut at --launch-timeout, and the attempts plus teardown at LAUNCH_TOTAL_S in all."""
    argv = [a for a in (sys.argv[1:] if argv is None else argv) if a != '--fallback-batch']
    modes = [args.mode or 'slab']
    if args.fallback_batch and modes[0] == 'slab':
        modes.append('batch')
    failures = []
    t_start = time.monotonic()
    for i, mode in enumerate(modes):
        left = LAUNCH_TOTAL_S - (time.monotonic() - t_start) - 15.0 * (len(modes) - i)           # 15 s of teardown per attempt still to come
        limit = min(args.launch_timeout, left)
        if limit < 25:
            failures.append(dict(mode=mode, status=124, note='not started: %.0f s left of the launcher\'s %.0f s' % (left, LAUNCH_TOTAL_S)))
            break
        a = argv + ([] if args.mode is not None else ['--mode', mode])
        log('bench: starting %d ranks (%s) as child processes, limit %.0f s ...' % (args.gpus, mode, limit))
        rc, lines, note = _spawn_ranks(args.gpus, a, limit, script)
        js = [l for l in lines if l.startswith('{')]
        if rc == 0 and js:
            line = js[-1]
            try:
                j = json.loads(line)
                j['launcher'] = dict(kind='self-launched child ranks (python bench.py --gpus %d)' % args.gpus, mode=mode, failed_attempts=failures,
                                     seconds=time.monotonic() - t_start)
                line = json.dumps(j)
            except ValueError:
                pass
            print(line, flush=True)
            return 0
        failures.append(dict(mode=mode, status=rc, note=note))
        log('bench: the %s attempt FAILED (status %s%s)%s' % (mode, rc, ', ' + note if note else '',
                                                               '' if i + 1 < len(modes) else ' -- no line; see the ranks\' output above'))
    return failures[-1]['status'] or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=30)       # the first ~25 steps after start-up run 3 % slower (tools/ramp_check.py)
    ap.add_argument('--n', type=int, default=1024)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--distinct', type=int, default=8, help='distinct synthetic grids generated on the host (tiled to --batch)')
    ap.add_argument('--stencil', type=int, default=5)
    ap.add_argument('--fast', action='store_true', help='precise=0: all-float32 transforms whatever the viscosity')
    ap.add_argument('--f64', action='store_true', help='precise=2: float64 forward transforms always (the default, precise=1, lets the library take the '
                                                        'all-float32 differenced mode when its error bound allows: it does at this configuration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--separate', action='store_true',
                    help='launch the FD and the spectral residual separately (3 launches) instead of the fused row pass (2 launches)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' only to rehearse the multi-process path on a one-GPU box)")
    ap.add_argument('--mode', choices=['batch', 'slab'], default=None,
                    help="N > 1 only.  slab (default): the SAME --batch grids are slab-decomposed by rows over the ranks -- halo "
                         "send/recv for the stencil, 2 all-to-alls per spectral evaluation (strong scaling); "
                         "batch: every rank owns --batch whole grids, no data-path collective (weak scaling)")
    ap.add_argument('--chunks', type=int, default=None, help='slab mode: batch chunks pipelined through the stages (default 1; the chunked pipeline is also timed as an extra, `pipelined`)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the `secondary` object (BASELINE configs 1, 2, 3, 5; ~20 s, N = 1 only)')
    ap.add_argument('--loopback', action='store_true', help='rehearsal on ONE GPU: --gpus 1 --mode slab --loopback runs the slab path with every message going '
                    'through a world-1 RCCL process group to the rank itself (device buffers, async collectives; not a scaling number)')
    ap.add_argument('--launch-timeout', type=float, default=240.0, help='self-launcher: seconds before the child ranks of ONE attempt are killed (all attempts + teardown: 540 s)')
    ap.add_argument('--fallback-batch', action='store_true', help='self-launcher: after a failed / hung slab attempt, run the ranks again in --mode batch (off by default: a slab failure exits non-zero)')
    ap.add_argument('--extras-timeout', type=float, default=150.0, help='N > 1: seconds the untimed extras after the headline (phases, chunked pipeline, batch splits) may take before the line is printed without them')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))                            # nothing has touched the GPU yet: `import torch` only
    if args.mode is None:
        args.mode = 'slab'
    if args.loopback and (args.gpus != 1 or args.backend != 'nccl'):
        raise SystemExit('--loopback is the one-GPU RCCL rehearsal: --gpus 1 --backend nccl')

    # RCCL peer-to-peer IPC on this pool needs the dmabuf mode (the host driver supports no legacy IPC handles: hipIpcGetMemHandle fails without
    # it); set here, before the first HIP call, so that self-launched children and torch.distributed.run ranks run in the SAME HSA mode
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    all_cores = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('bench: timing the NumPy oracle on all host cores (before the GPU is touched) ...')
        try:
            all_cores = cpu_baseline_all_cores(args.n)
        except Exception as e:                                     # noqa: BLE001 -- a reported extra, never fatal for the bench line
            log('bench: all-core CPU baseline skipped: %r' % (e,))
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == 'nccl' else local_rank % max(ndev, 1)     # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    if world > 1 or args.loopback:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.loopback:
            os.environ.setdefault('MASTER_PORT', str(_free_port())); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group('gloo')

    from nns import ops, _lib
    from nns.periodic import ResidualEngine
    n, B = args.n, args.batch
    prec = 0 if args.fast else (2 if args.f64 else 1)
    dt, nu, rho, L = 1e-3, 2 * np.pi / 1000, 1.0, 2 * np.pi
    if rank == 0:
        log('bench: device', _lib.device_info(), 'world', world)
        log('bench: generating %d distinct %dx%d grids on the host ...' % (min(args.distinct, B), n, n))
    slab = args.mode == 'slab' and (world > 1 or args.loopback)
    f = make_inputs(B, n, args.distinct, 1234 + (0 if slab else 1000 * rank), device)
    eng = ResidualEngine(n, n, dt, rho, nu, L, L, backend='spectral', precise=prec)
    if slab:
        from nns.slab import SlabResidual
        nloc = n // world
        f = [t[:, rank * nloc:(rank + 1) * nloc].contiguous() for t in f]          # this rank's rows of every grid
        sl = SlabResidual(n, n, dt, rho, nu, L, L, precise=prec, chunks=args.chunks, loopback=args.loopback)

        s_fd = tuple(torch.empty_like(f[0]) for _ in range(3))
        s_sp = tuple(torch.empty_like(f[0]) for _ in range(3))

        def step():           # per batch chunk: pack (+ halo rows), all-to-all, column pass, all-to-all, ONE fused row pass (5-point stencil) on the receive buffer
            sl.both(*f, stencil=args.stencil, out_fd=s_fd, out_spec=s_sp)
    else:
        out_fd = tuple(torch.empty_like(f[0]) for _ in range(3))
        out_sp = tuple(torch.empty_like(f[0]) for _ in range(3))

        def step():                           # FD + spectral residual of the same inputs: fused row pass with the 5-point stencil
            eng.both(*f, out_fd=out_fd, out_spec=out_sp, stencil=args.stencil, fused=not args.separate)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    if dist is not None:
        sync_all()               # (the first barriers of a process group can take tens of ms: two before the clock starts)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1 extras, measured after the timed region: what RCCL really saw, one un-pipelined evaluation phase by phase, and the
    # embarrassingly parallel alternative (every rank its own --batch grids)
    multi = None
    pts = float(B) * n * n
    value = (1 if slab else world) * pts * args.steps / elapsed          # slab: the ranks share ONE batch of grids
    watchdog = None
    if world > 1 or args.loopback:
        # The headline is measured.  What follows (phase timings, the chunked pipeline, the batch splits) are untimed EXTRAS over collectives that
        # no multi-GPU box has run before the driver's: if they do not finish within --extras-timeout, rank 0 prints the line with what it has and
        # every rank leaves (os._exit: a rank stuck inside a collective cannot be joined).
        import threading
        partial = dict(metric='grid-point residual-updates/sec at 1024^2 (FD 5-point + spectral residual on the same inputs)', value=value,
                       unit='residual-updates/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps,
                       higher_is_better=True, scaling='strong' if slab else 'weak', vs_baseline=None, dtype='f32', data='synthetic',
                       config=dict(workload='periodic-box NS residual, %dx%d, batch %d, FD %d-point + Fourier spectral' % (n, n, B, args.stencil),
                                   parallelism=('row-slab x%d' % world) if slab else 'batch-sharded x%d' % world),
                       roofline=None, cpu_baseline=None)

        def bail():
            if rank == 0:
                partial['extras'] = 'NOT COMPLETED within %.0f s: the line carries the timed headline only' % args.extras_timeout
                print(json.dumps(partial), flush=True)
            log('bench: rank %d: extras exceeded %.0f s -- leaving' % (rank, args.extras_timeout))
            os._exit(0)                        # every rank: the headline was measured and (rank 0) printed; a non-zero status would mark the whole run failed
        watchdog = threading.Timer(args.extras_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        ones = torch.ones(1, device=device if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(ones)
        multi = dict(transport=dict(backend=dist.get_backend(), library='RCCL (torch "nccl" on ROCm)' if args.backend == 'nccl' else
                                    'gloo: device buffers staged through the host, every collective blocking (rehearsal only)',
                                    device_buffers=args.backend == 'nccl', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')),
                     rccl_ranks=int(ones.item()) if args.backend == 'nccl' else 0, ranks_confirmed_by_all_reduce=int(ones.item()))
        if slab:
            # how long the HOST takes to enqueue one step (the per-chunk pipeline is ~10 launches / collectives per chunk from Python): if this
            # exceeds the device time of a step, the step is host-bound
            torch.cuda.synchronize()
            th0 = time.perf_counter()
            for _ in range(10):
                step()
            host_ms = 1e3 * (time.perf_counter() - th0) / 10
            torch.cuda.synchronize()
            multi['host_enqueue_ms_per_step'] = host_ms
            # the batch-chunk pipeline (collectives of chunk c+1 under the kernels of chunk c), timed like the headline: an extra until a multi-GPU
            # run has confirmed the interleaving (the headline runs --chunks, default 1)
            if sl._nchunks(B) == 1 and B >= 2:
                for _ in range(min(args.warmup, 5)):
                    sl.both(*f, stencil=args.stencil, chunks=2, out_fd=s_fd, out_spec=s_sp)
                sync_all()
                tp0 = time.perf_counter()
                for _ in range(args.steps):
                    sl.both(*f, stencil=args.stencil, chunks=2, out_fd=s_fd, out_spec=s_sp)
                sync_all()
                tpp = torch.tensor([time.perf_counter() - tp0], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
                dist.all_reduce(tpp, op=dist.ReduceOp.MAX)
                th0 = time.perf_counter()
                for _ in range(10):
                    sl.both(*f, stencil=args.stencil, chunks=2, out_fd=s_fd, out_spec=s_sp)
                hp = 1e3 * (time.perf_counter() - th0) / 10
                torch.cuda.synchronize()
                multi['pipelined'] = dict(chunks=2, value=pts * args.steps / float(tpp.item()), ms_per_step=1e3 * float(tpp.item()) / args.steps,
                                          host_enqueue_ms_per_step=hp, unit='residual-updates/s',
                                          note='the same step with the batch split into 2 chunks pipelined through pack / all-to-all / column pass / all-to-all / row pass')
            ph = slab_phases(sl, f, max(3, min(args.steps, 10)))
            keys = sorted(ph)
            t = torch.tensor([ph[k] for k in keys], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            multi['phases'] = dict(ms_max_over_ranks=dict(zip(keys, [float(x) for x in t.tolist()])), ms_rank0=ph,
                                   note='ONE un-pipelined evaluation (chunks=1) with a device synchronisation after every phase; the timed step '
                                        'pipelines %d batch chunks, so its time is less than the sum' % sl._nchunks(B))
            # the timed path's own numbers against the single-process kernels: rank 0 evaluates grid 0 .. 1 whole (it generated the full
            # inputs) and compares its rows of them with what the slab-decomposed, chunk-pipelined step just produced -- bitwise expected
            fd_s, sp_s = sl.both(*f, stencil=args.stencil)
            torch.cuda.synchronize()
            if rank == 0:
                full = make_inputs(min(B, 2), n, args.distinct, 1234, device)
                fd_1, sp_1 = eng.both(*full, stencil=args.stencil)
                nl = n // world
                diffs = [float((a[:full[0].shape[0]] - b[:, :nl]).abs().max()) for a, b in zip(list(fd_s) + list(sp_s), list(fd_1) + list(sp_1))]
                multi['slab_check'] = dict(compared='rank 0 rows of grids 0..%d: slab-decomposed step vs the single-process kernels on the whole grids' % (full[0].shape[0] - 1),
                                           max_abs_diff=max(diffs), bitwise_equal=all(x == 0.0 for x in diffs))
                del full, fd_1, sp_1
            del fd_s, sp_s
            a2a_bytes = 3.0 * B * (n // world) * n * 4 * (world - 1) / world       # bytes leaving this GPU per all-to-all
            multi['comm'] = dict(all_to_all_bytes_leaving_each_gpu=a2a_bytes, all_to_alls_per_step=2, halo_bytes_sent_each_gpu=2 * 3.0 * B * n * 4,
                                 chunks=sl._nchunks(B),
                                 all_to_all_GBs_per_gpu={k: a2a_bytes / (multi['phases']['ms_max_over_ranks'][k] * 1e-3) / 1e9
                                                         for k in ('all_to_all_1', 'all_to_all_2') if multi['phases']['ms_max_over_ranks'].get(k, 0) > 0})
            # batch-sharded alternative: every rank evaluates its OWN --batch whole grids, no data-path collective
            fb = make_inputs(B, n, args.distinct, 1234 + 1000 * rank, device)
            ob1 = tuple(torch.empty_like(fb[0]) for _ in range(3)); ob2 = tuple(torch.empty_like(fb[0]) for _ in range(3))
            for _ in range(min(args.warmup, 10)):
                eng.both(*fb, out_fd=ob1, out_spec=ob2, stencil=args.stencil)
            sync_all()
            tb0 = time.perf_counter()
            for _ in range(args.steps):
                eng.both(*fb, out_fd=ob1, out_spec=ob2, stencil=args.stencil)
            sync_all()
            tb = torch.tensor([time.perf_counter() - tb0], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            # ... and the SAME --batch grids split by whole grids over the ranks (strong scaling without a collective): rank r takes grids [r B/P, (r+1) B/P)
            g0, g1 = rank * B // world, (rank + 1) * B // world
            if g1 > g0:
                fs = [t[g0:g1].contiguous() for t in make_inputs(B, n, args.distinct, 1234, device)]
                os1 = tuple(torch.empty_like(fs[0]) for _ in range(3)); os2 = tuple(torch.empty_like(fs[0]) for _ in range(3))
            for _ in range(min(args.warmup, 10)):
                if g1 > g0:
                    eng.both(*fs, out_fd=os1, out_spec=os2, stencil=args.stencil)
            sync_all()
            ts0 = time.perf_counter()
            for _ in range(args.steps):
                if g1 > g0:
                    eng.both(*fs, out_fd=os1, out_spec=os2, stencil=args.stencil)
            sync_all()
            tsp = torch.tensor([time.perf_counter() - ts0], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(tsp, op=dist.ReduceOp.MAX)
            multi['batch_split'] = dict(value=float(B) * n * n * args.steps / float(tsp.item()), unit='residual-updates/s', scaling='strong',
                                        ms_per_step=1e3 * float(tsp.item()) / args.steps,
                                        note='the same %d grids as the headline, whole grids dealt to the ranks (%d each), no data-path collective' % (B, max(1, B // world)))
            multi['batch_sharded'] = dict(value=world * float(B) * n * n * args.steps / float(tb.item()), unit='residual-updates/s', scaling='weak',
                                          ms_per_step=1e3 * float(tb.item()) / args.steps,
                                          note='every rank its own %d grids, no data-path collective (not the headline: BASELINE config 4 is the slab decomposition)' % B)
            del fb, ob1, ob2

    # grid 0 of the timed step's outputs, for the oracle check next to the CPU baseline (the per-kernel timing below reuses the buffers)
    check_out = [t[0].cpu().numpy() for t in (*out_fd, *out_sp)] if (rank == 0 and not slab and world == 1 and not args.no_cpu_baseline and args.stencil == 5) else None
    # the same step with float64 forward transforms forced (precise=2), reported next to the headline for transparency (not `value`)
    alt = None
    if rank == 0 and not slab and world == 1 and prec == 1 and args.stencil == 5 and not args.separate:
        eng2 = ResidualEngine(n, n, dt, rho, nu, L, L, backend='spectral', precise=2)
        for _ in range(3):
            eng2.both(*f, out_fd=out_fd, out_spec=out_sp, stencil=5)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(10):
            eng2.both(*f, out_fd=out_fd, out_spec=out_sp, stencil=5)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - ta) / 10
        alt = dict(ms_per_step=1e3 * tb, value=float(B) * n * n / tb, steps=10)
    if watchdog is not None:
        watchdog.cancel()

    result = None
    if slab:                                   # per-kernel roofline is measured on whole grids: rebuild local full-size inputs
        f = make_inputs(B, n, args.distinct, 1234, device) if rank == 0 else f
        out_fd = tuple(torch.empty_like(f[0]) for _ in range(3))
        out_sp = tuple(torch.empty_like(f[0]) for _ in range(3))
    if rank == 0:
        # per-kernel durations, live, for the roofline object (same process, same inputs)
        iters = max(5, args.steps)
        fused = (not slab) and (not args.separate) and args.stencil == 5
        standalone = {
            'fd_residual': time_kernel(lambda: eng.fd(*f, stencil=args.stencil, out=out_fd), iters),
            'spec_xpass': time_kernel(lambda: ops.spec_residual_xpass(f[0], f[1], f[2], L, rho, nu, prec, out=out_sp), iters),
            'spec_ypass': time_kernel(lambda: ops.spec_residual_ypass_(*f, *out_sp, dt, L, rho, nu, prec), iters),
        }
        if fused:
            # the launches the timed step is made of, timed IN the step's own sequence (column pass, row pass, column pass, ...): HIP events
            # on the launch stream between the two launches of every step -- a row pass that follows the column pass finds part of the
            # partials in the Infinity Cache, one that follows another row pass (a back-to-back loop of the same launch) does not
            def rowpass():
                ops.residual_both(*f, dt, L, L, rho, nu, prec, out_fd=out_fd, out_spec=out_sp, rowpass_only=True)

            def xpass():
                ops.spec_residual_xpass(f[0], f[1], f[2], L, rho, nu, prec, out=out_sp)
            for _ in range(5):
                xpass(); rowpass()
            torch.cuda.synchronize()
            evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(iters)]
            for e0, e1, e2 in evs:
                e0.record(); xpass(); e1.record(); rowpass(); e2.record()
            torch.cuda.synchronize()
            kt = {'spec_xpass': sum(e0.elapsed_time(e1) for e0, e1, _ in evs) / iters,
                  'both_rowpass': sum(e1.elapsed_time(e2) for _, e1, e2 in evs) / iters}
            standalone['both_rowpass_back_to_back'] = time_kernel(rowpass, iters)
        else:
            kt = dict(standalone)
        # this box's own streaming ceiling, measured in this run: a device-to-device copy of one 268 MB field (1 read + 1 write stream).  The pool's
        # boxes differ by ~10 % in every HBM-bound kernel; tools/streams_bench.hip (profiles/r03_streams_bench.txt) shows a pure copy with the row
        # pass's 8 + 6 streams within 5 % of this figure, and the row pass at 96-100 % of that copy.
        box_copy = None
        try:
            src, dst = f[0], torch.empty_like(f[0])
            for _ in range(3):
                dst.copy_(src)
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(10):
                dst.copy_(src)
            c1.record(); torch.cuda.synchronize()
            box_copy = 2.0 * src.numel() * src.element_size() * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del dst
        except Exception:
            box_copy = None
        dom = max(kt, key=kt.get)
        alg_bytes = BYTES_PER_PT[dom] * pts
        achieved = alg_bytes / (kt[dom] * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters are builder evidence collected in a SEPARATE rocprofv3 --pmc pass
        # (tools/profile_round.sh + tools/hbm_traffic.py), not measured in this run: the record says where they come from.
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom)
                traffic_source = dict(file='profiles/hbm_traffic.json', measured_in_this_run=False, **tj.get('_source', {}))
            except Exception:
                traffic = None
        # the whole step against the roofline, two ways:
        #   two_pass   = what the separable two-launch design must move: column pass 24 + fused row pass 56 = 80 B/pt
        #   compulsory = the metric's unit alone: 5 input fields read once, 6 residual fields written once = 44 B/pt
        step_s = elapsed / args.steps
        step_obj = dict(ms=1e3 * step_s, bytes_per_pt_two_pass=STEP_BYTES_TWO_PASS, bytes_per_pt_compulsory=STEP_BYTES_COMPULSORY,
                        achieved_GBs=dict(two_pass=STEP_BYTES_TWO_PASS * pts / step_s / 1e9, compulsory=STEP_BYTES_COMPULSORY * pts / step_s / 1e9),
                        frac=dict(two_pass=STEP_BYTES_TWO_PASS * pts / step_s / 1e9 / HBM_PEAK_GBS,
                                  compulsory=STEP_BYTES_COMPULSORY * pts / step_s / 1e9 / HBM_PEAK_GBS)) if fused and not slab else None
        roofline = dict(bound='hbm', kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                        traffic=traffic, traffic_source=traffic_source, step=step_obj,
                        box_copy_GBs=box_copy, frac_of_box_copy=(achieved / box_copy if box_copy else None),
                        algorithmic_bytes_per_launch=alg_bytes, avg_launch_ms=kt[dom],
                        all_kernels={k: dict(avg_launch_ms=v, bytes_per_pt=BYTES_PER_PT[k],
                                             achieved_GBs=BYTES_PER_PT[k] * pts / (v * 1e-3) / 1e9) for k, v in kt.items()},
                        step_launches=list(kt),
                        standalone_kernels={k: dict(avg_launch_ms=v, bytes_per_pt=BYTES_PER_PT[k],
                                                    achieved_GBs=BYTES_PER_PT[k] * pts / (v * 1e-3) / 1e9) for k, v in standalone.items()})
        result = dict(metric='grid-point residual-updates/sec at 1024^2 (FD 5-point + spectral residual on the same inputs)',
                      parity_note='operator defined by oracle/periodic.py (the reference has no periodic residual: SURVEY 8 row a17), pinned analytically; '
                                  'tests hold the kernels to 1e-5 rel-L2 of that float64 oracle',
                      value=value, unit='residual-updates/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                      ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling='strong' if slab else 'weak', vs_baseline=None,
                      scaling_note=('the SAME %d grids on all %d GPUs, each grid slab-decomposed by rows (BASELINE config 4): every evaluation sends 24 (P-1)/P B/pt '
                                    'over xGMI (u, v, p out, three partials back) against the 80 B/pt its kernels move in HBM at ~5 TB/s; the links are point-to-point '
                                    '(P = 2: ONE link carries 12 B/pt per direction), so the decomposition is link-bound at every P and slower than one GPU at small P. '
                                    '`batch_split` (same grids, whole grids per rank, strong) and `batch_sharded` (weak) are the no-collective alternatives, measured in '
                                    'the same run' % (B, world)) if slab else None,
                      dtype='f32' if prec < 2 else 'f32 fields; f64 forward FFT + f32 inverse FFT; f32 stencil',
                      precision=dict(precise=prec, note='precise=1: the library takes all-float32 transforms on forward-differenced lines while the viscous '
                                     'amplification nu pi N/(sqrt(3) L) <= 8 (1.86 here), float64 forward transforms otherwise; rel-L2 against the float64 '
                                     'oracle in cpu_baseline.oracle_check', float64_forward_forced=alt),
                      data='synthetic',
                      config=dict(workload='periodic-box NS residual, %dx%d, batch %d grids per GPU, FD %d-point + Fourier spectral%s'
                                           % (n, n, B, args.stencil, ' (fused row pass: 2 launches)' if fused else
                                              (' (per rank and batch chunk: pack, all-to-all, column pass, all-to-all, unpack, fused row pass)' if slab else ' (separate launches)')),
                                  grid=[n, n], batch_per_gpu=B, global_batch=B if slab else B * world,
                                  parallelism=('row-slab x%d: the same %d grids on all ranks; halo send/recv + 2 all-to-all transposes per evaluation over %s'
                                               % (world, B, 'RCCL' if args.backend == 'nccl' else 'gloo (host-staged rehearsal)')) if slab
                                  else 'batch-sharded x%d (no data-path collective)' % world,
                                  inputs='Taylor-Green t=0.1 + band-limited noise (seeds 1234 ...), nu=2pi/1000, dt=1e-3, resident in HBM; %d DISTINCT grids '
                                         'generated on the host and tiled to the batch of %d (timing is data-independent; all %d grids are separate arrays in HBM)'
                                         % (min(args.distinct, B), B, B)),
                      roofline=roofline)
        if multi is not None:
            result.update(multi)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_secondary:
            log('bench: secondary configs (BASELINE configs 1, 2, 3, 5) ...')
            try:
                import bench_configs
                result['secondary'] = bench_configs.secondary(cpu=not args.no_cpu_baseline)
            except Exception as e:                                 # noqa: BLE001 -- an extra object: never lose the headline line over it
                result['secondary'] = dict(error=repr(e))
        if not args.no_cpu_baseline:
            log('bench: timing the NumPy oracle on the host (bounded sample) ...')
            # N > 1: the same single-core sample on rank 0's host cores, shorter (the other ranks wait at the barrier below); the all-core row
            # needs a fork before the GPU is touched and stays an N = 1 extra
            result['cpu_baseline'] = cpu_baseline(n, budget_s=20.0 if world == 1 else 10.0, check=check_out)
            result['cpu_baseline']['all_cores'] = all_cores        # extra row: one oracle process per usable host core
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
