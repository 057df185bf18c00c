"""Periodic-box incompressible Navier-Stokes residual engine (the north-star hot path).

    r_u   = (u - u_prev)/dt + u u_x + v u_y + p_x/rho - nu lap u
    r_v   = (v - v_prev)/dt + u v_x + v v_y + p_y/rho - nu lap v
    r_div = u_x + v_y

on batches of [B, nx, ny] float32 fields resident in HBM, with two derivative back-ends:
  * 'fd5' / 'fd9'  -- 2nd-order central differences, 5- or 9-point Laplacian (csrc/residual_kernels.hip)
  * 'spectral'     -- Fourier derivatives via LDS-resident FFTs (csrc/spectral_kernels.hip)
The reference has no such operator (motivation
src/neural_spectral/derivations/derivation.tex:25-59); oracle/periodic.py defines it and the
tests pin both back-ends to it (1e-5 rel-L2 in float32).  Axis 0 = x, axis 1 = y.

``precise`` (spectral back-end; include/nns.h): False / 0 = all-float32 transforms of forward-differenced lines, True / 1 = the
library picks that mode while its viscous amplification nu pi N / (sqrt(3) L) stays <= 8 and float64 forward transforms
otherwise, 2 = float64 forward transforms always.
"""
import math

import torch

from . import ops


class ResidualEngine(object):
    def __init__(self, nx, ny, dt, rho, nu, Lx=2 * math.pi, Ly=2 * math.pi, backend='spectral', precise=True):
        if backend not in ('fd5', 'fd9', 'spectral'):
            raise ValueError("backend must be 'fd5', 'fd9' or 'spectral'")
        self.nx, self.ny, self.dt, self.rho, self.nu = nx, ny, dt, rho, nu
        self.Lx, self.Ly = Lx, Ly
        self.dx, self.dy = Lx / nx, Ly / ny
        self.backend, self.precise = backend, precise

    def fd(self, u, v, p, u_prev, v_prev, stencil=5, out=None):
        return ops.fd_residual(u, v, p, u_prev, v_prev, self.dt, self.dx, self.dy, self.rho, self.nu, stencil, out)

    def spectral(self, u, v, p, u_prev, v_prev, out=None):
        return ops.spec_residual(u, v, p, u_prev, v_prev, self.dt, self.Lx, self.Ly, self.rho, self.nu, self.precise, out)

    def __call__(self, u, v, p, u_prev, v_prev, out=None):
        if self.backend == 'spectral':
            return self.spectral(u, v, p, u_prev, v_prev, out)
        return self.fd(u, v, p, u_prev, v_prev, 5 if self.backend == 'fd5' else 9, out)

    def differentiable(self, u, v, p, u_prev, v_prev):
        """The residual as an autograd node (inputs may require grad): forward and backward are both HIP kernels
        (the backward applies the adjoint operators, see oracle/periodic.py: residual_vjp)."""
        if self.backend == 'spectral':
            return ops.SpecResidualFn.apply(u, v, p, u_prev, v_prev, self.dt, self.Lx, self.Ly, self.rho, self.nu, self.precise)
        return ops.FdResidualFn.apply(u, v, p, u_prev, v_prev, self.dt, self.dx, self.dy, self.rho, self.nu, 5 if self.backend == 'fd5' else 9)

    def residual_spec(self):
        """(kind, constants) of this engine's residual as ops.PinnHeadFn takes it."""
        if self.backend == 'spectral':
            return 'spectral', (self.dt, self.Lx, self.Ly, self.rho, self.nu, self.precise)
        return 'fd', (self.dt, self.dx, self.dy, self.rho, self.nu, 5 if self.backend == 'fd5' else 9)

    def physics_loss(self, u, v, p, u_prev, v_prev, w_div=1.0):
        """Mean-square momentum + divergence residual: the physics-informed loss term of SURVEY.md section 8 (f) rank 2
        (hypothesis: src/neural_spectral/derivations/derivation.tex:25-34)."""
        r_u, r_v, r_d = self.differentiable(u, v, p, u_prev, v_prev)
        return (r_u * r_u).mean() + (r_v * r_v).mean() + w_div * (r_d * r_d).mean()

    def both(self, u, v, p, u_prev, v_prev, out_fd=None, out_spec=None, stencil=5, fused=True):
        """The 'stencil + spectral residual' of BASELINE.json: both back-ends on the same inputs.  With the 5-point stencil
        and float32 fields this is nns_residual_both_f32: the spectral column pass and ONE row pass that also
        evaluates the stencil (the inputs cross HBM once less); otherwise, or with fused=False, the two back-ends are
        launched separately.  Same results either way, to rounding."""
        if fused and stencil == 5 and u.dtype == torch.float32:
            return ops.residual_both(u, v, p, u_prev, v_prev, self.dt, self.Lx, self.Ly, self.rho, self.nu, self.precise,
                                     out_fd=out_fd, out_spec=out_spec)
        return (self.fd(u, v, p, u_prev, v_prev, stencil, out_fd), self.spectral(u, v, p, u_prev, v_prev, out_spec))
