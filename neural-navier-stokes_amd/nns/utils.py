"""Mirror of the data-preparation helper of the reference's ``src/utils.py`` that is on the scope table (SURVEY.md section 8 row f4:
``spatial_coarsen``) on the HIP library.  The file's training-loop leftovers (AverageMeter, save_checkpoint, mean_squared_error) are dead code
in the reference (out of scope) and are not carried here; the drivers' own running-average helper lives in
nns/neural_spectral/spectral_ode.py.

spatial_coarsen runs on the GPU (``nns_coarsen_*``: one launch for u, v, p; numpy.mean's pairwise add order is
reproduced in the kernel, so float64 results are bit-identical to the reference's).  No CPU fallback.
"""
import numpy as np
import torch

from . import ops


def numpy_to_torch(array, device):
    """float32 tensor on `device` from a NumPy array (src/utils.py:9-10)."""
    return torch.from_numpy(array).float().to(device)


def spatial_coarsen(X, Y, u_seq, v_seq, p_seq, agg_x=4, agg_y=4):
    """Average the [T, nx, ny] sequences over agg_x x agg_y blocks (src/utils.py:13-60).

    Returns (new_X, new_Y, new_u_seq, new_v_seq, new_p_seq) like the reference, including its conventions: the new
    meshgrid is numpy's default 'xy' one over linspace(0, 2, .) (so [ny/agg_y, nx/agg_x]), and the column loop covers
    ny // agg_x blocks -- cells beyond stay 0, and agg combinations that overrun the coarse array raise IndexError as
    the reference does.  NumPy in -> NumPy float64 out (computed in float64 on the device); device tensors in
    (float32 / float64) -> device tensors out, no host copy."""
    nx, ny = X.shape[0], X.shape[1]
    assert nx % agg_x == 0
    assert ny % agg_y == 0
    cny = ny // agg_y
    jfill = ny // agg_x
    if jfill > cny:
        raise IndexError("index %d is out of bounds for axis 2 with size %d" % (cny, cny))
    new_X, new_Y = np.meshgrid(np.linspace(0, 2, nx // agg_x), np.linspace(0, 2, cny))
    as_numpy = not torch.is_tensor(u_seq)
    if as_numpy:
        f = [torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device='cuda') for a in (u_seq, v_seq, p_seq)]
    else:
        f = [a.contiguous() for a in (u_seq, v_seq, p_seq)]
    cu, cv, cp = ops.coarsen(f[0], f[1], f[2], agg_x, agg_y, jfill)
    if as_numpy:
        cu, cv, cp = (a.cpu().numpy() for a in (cu, cv, cp))
    return new_X, new_Y, cu, cv, cp
