"""The reference's driver scripts run unchanged against this package (PYTHONPATH = neural-navier-stokes_amd):
on-disk formats of SURVEY.md section 8 row f1 -- data_{method}.npz (u, v, p float64 [nt, nx, ny]),
checkpoint.pth.tar (model_state_dict, optimizer_state_dict, config, losses, penalties), extrapolation.npy."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, load_golden, rel_l2

pytestmark = pytest.mark.gpu


def run(script, args, cwd):
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, 'src', script)] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_chorin_fd_driver_then_neural_training_driver(tmp_path, gpu_device):
    d = str(tmp_path)
    run('chorin_fd/simulate.py', ['--nt', '5', '--nit', '50', '--nx', '64', '--ny', '64', '--nu', '0.02', '--method', 'explicit'], d)
    data = np.load(os.path.join(d, 'data_explicit.npz'))
    assert sorted(data.files) == ['p', 'u', 'v'] and data['u'].shape == (5, 64, 64) and data['u'].dtype == np.float64
    g = load_golden('chorin_fd_cavity_64_explicit_nu0.02.npz')                 # BASELINE config 1 through the driver
    assert rel_l2(data['u'][[0, -1]], g['u']) < 1e-9 and rel_l2(data['p'][[0, -1]], g['p']) < 1e-9
    run('direct_fd/simulate.py', ['--nt', '3', '--nx', '16', '--ny', '16'], d)
    assert np.load(os.path.join(d, 'data.npz'))['p'].shape == (3, 16, 16)
    for script in ('neural_spectral/spectral_ode.py', 'neural_spectral/spectral_ode2.py'):
        out = os.path.join(d, 'ck_' + os.path.basename(script)[:-3])
        run(script, ['--npz-path', os.path.join(d, 'data_explicit.npz'), '--out-dir', out, '--n-iters', '20', '--n-coeffs', '4'], d)
        ck = torch.load(os.path.join(out + '_4', 'checkpoint.pth.tar'), weights_only=False)
        assert sorted(ck.keys()) == ['config', 'losses', 'model_state_dict', 'optimizer_state_dict', 'penalties']
        assert ck['losses'].shape == (20,) and ck['losses'][-1] < ck['losses'][0]       # it trains
        assert ck['config'].n_coeffs == 4
        ex = np.load(os.path.join(out + '_4', 'extrapolation.npy'))
        assert ex.shape == (5, 3, 64, 64) and ex.dtype == np.float32
