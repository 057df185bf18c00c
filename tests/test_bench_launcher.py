"""`python bench.py --gpus N` (N > 1) outside a launcher starts its own child ranks (bench.self_launch): the rendezvous
environment every rank gets, the relay of rank 0's JSON line, a failed slab attempt surfacing as a non-zero exit (the slab -> batch
fallback only behind --fallback-batch, with the failure on record), and the bounded wait -- one attempt and all attempts together.  CPU only: the ranks here are a stub script (a gloo all-reduce instead of the residual kernels) -- the launcher
is what is under test, not the benchmark body."""
import argparse
import json
import os
import sys
import textwrap

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

STUB = textwrap.dedent('''
    import json, os, sys, time
    import torch, torch.distributed as dist
    mode = sys.argv[sys.argv.index('--mode') + 1] if '--mode' in sys.argv else None
    behaviour = os.environ.get('STUB_BEHAVIOUR', 'ok')
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert int(os.environ['LOCAL_RANK']) == rank and os.environ['MASTER_ADDR'] == '127.0.0.1'
    if behaviour == 'slab_fails' and mode == 'slab' and rank == world - 1:
        sys.exit(7)
    if behaviour == 'hang':
        time.sleep(600)
    dist.init_process_group('gloo')
    t = torch.ones(1)
    dist.all_reduce(t)
    if rank == 0:
        print('some log line on stdout')
        print(json.dumps(dict(value=float(t.item()), mode=mode, n_gpus=world)), flush=True)
    else:
        print('rank %d says hello on stdout (must not reach the launcher\\'s stdout)' % rank)
    dist.destroy_process_group()
''')


@pytest.fixture
def stub(tmp_path):
    p = tmp_path / 'stub_rank.py'
    p.write_text(STUB)
    return str(p)


def _args(gpus, mode=None, timeout=120.0, fallback_batch=False):
    return argparse.Namespace(gpus=gpus, mode=mode, launch_timeout=timeout, fallback_batch=fallback_batch)


@pytest.mark.parametrize('world', [2, 3])
def test_self_launch_relays_rank0_json(world, stub, capfd, monkeypatch):
    monkeypatch.setenv('STUB_BEHAVIOUR', 'ok')
    rc = bench.self_launch(_args(world), argv=['--gpus', str(world)], script=stub)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1                              # ONE line on stdout: rank 0's JSON
    j = json.loads(out[0])
    assert j['value'] == world and j['n_gpus'] == world and j['mode'] == 'slab'
    assert j['launcher']['mode'] == 'slab' and j['launcher']['failed_attempts'] == []


def test_self_launch_failed_slab_attempt_is_a_failure(stub, capfd, monkeypatch):
    """a slab run that crashes must not be replaced silently by a weak-scaling batch number -- no line, the rank's status."""
    monkeypatch.setenv('STUB_BEHAVIOUR', 'slab_fails')
    rc = bench.self_launch(_args(2), argv=['--gpus', '2'], script=stub)
    cap = capfd.readouterr()
    assert rc == 7 and cap.out.strip() == '' and 'FAILED' in cap.err


def test_self_launch_falls_back_to_batch_only_when_asked_and_says_so(stub, capfd, monkeypatch):
    monkeypatch.setenv('STUB_BEHAVIOUR', 'slab_fails')
    rc = bench.self_launch(_args(2, fallback_batch=True), argv=['--gpus', '2', '--fallback-batch'], script=stub)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1
    j = json.loads(out[0])
    assert j['mode'] == 'batch' and j['launcher']['mode'] == 'batch'
    assert j['launcher']['failed_attempts'] == [dict(mode='slab', status=7, note='rank 1 exited with status 7')]


def test_self_launch_total_time_is_bounded(stub, capfd, monkeypatch):
    """Both attempts plus teardown fit the launcher's total (540 s against the driver's 600 s bench limit): with the total shrunk to 60 s, a
    hung slab attempt is cut at 60 - 2 x 15 = 30 s and the batch attempt, for which less than 30 s would remain, is not started."""
    import time
    assert bench.LAUNCH_TOTAL_S <= 540.0
    monkeypatch.setattr(bench, 'LAUNCH_TOTAL_S', 60.0)
    monkeypatch.setenv('STUB_BEHAVIOUR', 'hang')
    t0 = time.monotonic()
    rc = bench.self_launch(_args(2, timeout=240.0, fallback_batch=True), argv=['--gpus', '2', '--fallback-batch'], script=stub)
    el = time.monotonic() - t0
    assert rc == 124 and capfd.readouterr().out.strip() == ''
    assert 28.0 < el < 50.0, el


def test_default_launch_timeouts_fit_the_driver_clock():
    """The default --launch-timeout of two attempts, their teardown (15 s each) and start-up fit 540 s."""
    import re
    src = open(os.path.join(ROOT, 'bench.py')).read()
    m = re.search(r"'--launch-timeout', type=float, default=([0-9.]+)", src)
    assert m and 2 * (float(m.group(1)) + 15.0) <= bench.LAUNCH_TOTAL_S <= 540.0


def test_self_launch_explicit_mode_has_no_fallback(stub, capfd, monkeypatch):
    monkeypatch.setenv('STUB_BEHAVIOUR', 'slab_fails')
    rc = bench.self_launch(_args(2, mode='slab'), argv=['--gpus', '2', '--mode', 'slab'], script=stub)
    assert rc == 7 and capfd.readouterr().out.strip() == ''


def test_self_launch_bounded_wait_kills_its_children(stub, capfd, monkeypatch):
    monkeypatch.setenv('STUB_BEHAVIOUR', 'hang')
    rc = bench.self_launch(_args(2, mode='batch', timeout=8.0), argv=['--gpus', '2', '--mode', 'batch'], script=stub)
    assert rc == 124 and capfd.readouterr().out.strip() == ''


def test_bench_under_a_launcher_is_a_rank_not_a_launcher(monkeypatch):
    """WORLD_SIZE in the environment (torch.distributed.run): bench.py must not start children of its own, and a world size that
    does not match --gpus is an error before anything touches the GPU."""
    monkeypatch.setenv('WORLD_SIZE', '4')
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    monkeypatch.setattr(bench, 'self_launch', lambda *a, **k: pytest.fail('launcher used under a launcher'))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=4 does not match --gpus 2' in str(e.value)
