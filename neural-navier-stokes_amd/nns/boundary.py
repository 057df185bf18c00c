"""Boundary conditions: mirror of the reference's ``src/boundary.py`` (same class names, ctor
arguments, attributes, asserts and in-place ``apply`` semantics).

``apply(A)`` runs the HIP kernel ``nns_bc_apply_*`` (csrc/fd_kernels.hip):
  * A is a CUDA/HIP torch tensor ([nx, ny] or [batch, nx, ny], float32/float64): in place on the device;
  * A is a NumPy array: copied to the device, updated there, copied back INTO ``A`` (the
    reference mutates and returns the same object -- so do we).
Sides (src/boundary.py:39-46): left = A[0, :], right = A[-1, :], bottom = A[:, 0], top = A[:, -1].
"""
import numpy as np
import torch

from . import ops
from ._util import default_device


class BaseBoundaryCondition(object):
    """src/boundary.py:1-26"""

    def __init__(self, value, boundary, dx, dy):
        super().__init__()
        assert isinstance(boundary, str)
        assert isinstance(dx, float)
        assert isinstance(dy, float)
        assert boundary in ['left', 'right', 'bottom', 'top']
        self.value = value
        self.boundary = boundary
        self.dx, self.dy = dx, dy

    def apply(self, A):
        raise NotImplementedError


def apply_list(A, bcs):
    """Apply a list of BCs in list order with ONE kernel launch (corner semantics preserved)."""
    if isinstance(A, torch.Tensor):
        ops.bc_apply_(A, bcs)
        return A
    dt = A.dtype if A.dtype in (np.float32, np.float64) else np.dtype('float64')
    d = torch.as_tensor(np.ascontiguousarray(A, dtype=dt), device=default_device())
    ops.bc_apply_(d, bcs)
    A[...] = d.cpu().numpy()
    return A


class _Applies(BaseBoundaryCondition):
    def apply(self, A):
        return apply_list(A, [self])


class DirichletBoundaryCondition(_Applies):
    """src/boundary.py:29-48"""

    def __init__(self, value, boundary, dx, dy):
        super().__init__(value, boundary, dx, dy)
        self.type = 'dirichlet'


class NeumannBoundaryCondition(_Applies):
    """src/boundary.py:51-86 (first-order one-sided difference)"""

    def __init__(self, value, boundary, dx, dy):
        super().__init__(value, boundary, dx, dy)
        self.type = 'neumann'
