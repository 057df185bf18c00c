"""spatial_coarsen (reference src/utils.py:13-60): oracle vs the reference's goldens (CPU), HIP kernel vs both (GPU)."""
import numpy as np
import pytest

from conftest import load_golden

CASES = 'abcdefgh'


def case(d, tag):
    T, nx, ny, ax, ay = (int(x) for x in d[tag + '_shape'])
    X, Y = np.meshgrid(np.linspace(0, 2, nx), np.linspace(0, 2, ny), indexing='ij')
    return X, Y, d[tag + '_u'], d[tag + '_v'], d[tag + '_p'], ax, ay


@pytest.mark.parametrize('tag', CASES)
def test_oracle_matches_reference_bitwise(tag):
    from oracle.utils import spatial_coarsen
    d = load_golden('utils_coarsen.npz')
    X, Y, u, v, p, ax, ay = case(d, tag)
    got = spatial_coarsen(X, Y, u, v, p, agg_x=ax, agg_y=ay)
    for g, name in zip(got, ('X', 'Y', 'cu', 'cv', 'cp')):
        assert g.shape == d[tag + '_' + name].shape and np.array_equal(g, d[tag + '_' + name]), (tag, name)


def test_oracle_overrun_raises_like_the_reference():
    from oracle.utils import spatial_coarsen
    d = load_golden('utils_coarsen.npz')
    assert str(d['overrun_error']) == 'IndexError'
    z = np.zeros((1, 8, 8))
    X, Y = np.meshgrid(np.linspace(0, 2, 8), np.linspace(0, 2, 8), indexing='ij')
    with pytest.raises(IndexError):
        spatial_coarsen(X, Y, z, z, z, agg_x=2, agg_y=4)
    with pytest.raises(AssertionError):
        spatial_coarsen(X, Y, z, z, z, agg_x=3, agg_y=2)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', CASES)
def test_hip_matches_reference_bitwise(tag, gpu_device):
    """float64 through the mirror: the kernel follows numpy.mean's pairwise order, so equality is exact; through the
    reference's import path (src.utils)."""
    from src.utils import spatial_coarsen
    d = load_golden('utils_coarsen.npz')
    X, Y, u, v, p, ax, ay = case(d, tag)
    got = spatial_coarsen(X, Y, u, v, p, agg_x=ax, agg_y=ay)
    for g, name in zip(got, ('X', 'Y', 'cu', 'cv', 'cp')):
        assert g.dtype == np.float64 and g.shape == d[tag + '_' + name].shape
        assert np.array_equal(g, d[tag + '_' + name]), (tag, name)


@pytest.mark.gpu
def test_hip_errors_like_the_reference(gpu_device):
    from nns.utils import spatial_coarsen
    z = np.zeros((1, 8, 8))
    X, Y = np.meshgrid(np.linspace(0, 2, 8), np.linspace(0, 2, 8), indexing='ij')
    with pytest.raises(IndexError):
        spatial_coarsen(X, Y, z, z, z, agg_x=2, agg_y=4)
    with pytest.raises(AssertionError):
        spatial_coarsen(X, Y, z, z, z, agg_x=3, agg_y=2)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(5, 256, 384, 4, 4), (2, 512, 512, 32, 16), (1, 1024, 1024, 64, 64), (3, 300, 1001, 1, 1)])
def test_hip_vs_oracle_large_and_deep_recursion(shape, gpu_device):
    """Sizes beyond the goldens (ragged launch tails, blocks of 512 and 4096 cells = 2 and 5 levels of the pairwise
    recursion, the 1x1 identity): float64 bitwise vs the oracle, float32 device tensors to 1e-6."""
    import torch
    from nns import ops
    from oracle.utils import spatial_coarsen as oracle_coarsen
    T, nx, ny, ax, ay = shape
    rng = np.random.default_rng(3)
    f = [rng.standard_normal((T, nx, ny)) for _ in range(3)]
    X = np.zeros((nx, ny))
    ref = oracle_coarsen(X, X, *f, agg_x=ax, agg_y=ay)[2:]
    got = ops.coarsen(*[torch.as_tensor(a, device='cuda') for a in f], ax, ay, jfill=ny // ax)     # the reference's column count (:49)
    for g, r in zip(got, ref):
        assert np.array_equal(g.cpu().numpy(), r)
    got32 = ops.coarsen(*[torch.as_tensor(a.astype(np.float32), device='cuda') for a in f], ax, ay, jfill=ny // ax)
    for g, r in zip(got32, ref):
        assert g.dtype == torch.float32 and np.abs(g.cpu().numpy() - r).max() < 1e-6
    if ax == 1 and ay == 1:
        assert np.array_equal(got[0].cpu().numpy(), f[0])


@pytest.mark.gpu
def test_hip_device_tensors_stay_on_device(gpu_device):
    import torch
    from nns.utils import spatial_coarsen
    u = torch.randn(2, 16, 16, device='cuda')
    X = np.zeros((16, 16))
    _, _, cu, cv, cp = spatial_coarsen(X, X, u, u * 2, u * 3, agg_x=4, agg_y=4)
    assert cu.is_cuda and cu.shape == (2, 4, 4)
    assert torch.allclose(cu, u.reshape(2, 4, 4, 4, 4).mean(dim=(2, 4)), atol=1e-6)
    assert torch.allclose(cp, 3 * cu, atol=1e-6)
