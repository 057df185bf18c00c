"""Oracle (test infrastructure only): boundary-condition application.

Restates ``src/boundary.py:34-48`` (Dirichlet) and ``src/boundary.py:56-86`` (Neumann) of the
reference as one pure function over a (kind, side, value, dx, dy) tuple, so that the HIP
``nns_bc_apply_*`` kernel and the fused in-kernel BC epilogues can be checked against it.

Sides (reference naming, src/boundary.py:39-46): 'left' = A[0, :], 'right' = A[-1, :],
'bottom' = A[:, 0], 'top' = A[:, -1].  List order decides corners (later entries win).
"""
import numpy as np

SIDES = ('left', 'right', 'bottom', 'top')
KINDS = ('dirichlet', 'neumann')


def apply_bc(A, kind, side, value, dx, dy):
    """In place on the trailing two axes of ``A`` ([..., nx, ny]); returns ``A``."""
    assert kind in KINDS and side in SIDES
    if kind == 'dirichlet':                       # src/boundary.py:39-46
        if side == 'left':
            A[..., 0, :] = value
        elif side == 'right':
            A[..., -1, :] = value
        elif side == 'bottom':
            A[..., :, 0] = value
        else:
            A[..., :, -1] = value
    else:                                         # src/boundary.py:73-84
        if side == 'left':
            A[..., 0, :] = A[..., 1, :] - dx * value
        elif side == 'right':
            A[..., -1, :] = A[..., -2, :] + dx * value
        elif side == 'bottom':
            A[..., :, 0] = A[..., :, 1] - dy * value
        else:
            A[..., :, -1] = A[..., :, -2] + dy * value
    return A


def apply_bc_list(A, bcs):
    """``bcs`` = iterable of (kind, side, value, dx, dy); applied in list order."""
    for (kind, side, value, dx, dy) in bcs:
        apply_bc(A, kind, side, value, dx, dy)
    return A


def cavity_bcs(dx, dy, lid=1.0):
    """The three BC lists of the reference drivers (src/chorin_fd/simulate.py:296-315,
    src/direct_fd/simulate.py:166-185): u (lid on 'right'), v (all zero), p."""
    u_bc = [('dirichlet', 'left', 0.0, dx, dy), ('dirichlet', 'right', lid, dx, dy),
            ('dirichlet', 'top', 0.0, dx, dy), ('dirichlet', 'bottom', 0.0, dx, dy)]
    v_bc = [('dirichlet', 'left', 0.0, dx, dy), ('dirichlet', 'right', 0.0, dx, dy),
            ('dirichlet', 'top', 0.0, dx, dy), ('dirichlet', 'bottom', 0.0, dx, dy)]
    p_bc = [('dirichlet', 'top', 0.0, dx, dy), ('neumann', 'bottom', 0.0, dx, dy),
            ('neumann', 'left', 0.0, dx, dy), ('neumann', 'right', 0.0, dx, dy)]
    return u_bc, v_bc, p_bc
