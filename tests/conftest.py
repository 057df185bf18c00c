import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'neural-navier-stokes_amd')
GOLD = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

KINDS = ('dirichlet', 'neumann')
SIDES = ('left', 'right', 'bottom', 'top')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLD, name))


def unpack_bcs(g, prefix):
    """Fixture arrays -> list of (kind, side, value, dx, dy) tuples (oracle form)."""
    return [(KINDS[int(k)], SIDES[int(s)], float(v), float(dx), float(dy))
            for k, s, v, dx, dy in zip(g[prefix + '_kind'], g[prefix + '_side'], g[prefix + '_value'],
                                       g[prefix + '_dx'], g[prefix + '_dy'])]


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (d if d > 0 else 1.0)


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device('cuda:0')
