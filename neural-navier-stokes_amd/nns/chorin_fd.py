"""Chorin projection, finite differences: mirror of the reference's
``src/chorin_fd/simulate.py`` ``NavierStokesSystem`` (same ctor, ``step``, ``simulate`` and
private operator methods), running on hand-written HIP kernels (csrc/fd_kernels.hip,
csrc/sor_kernels.hip) with fields resident on the GPU.

Reference semantics kept (file:line in src/chorin_fd/simulate.py):
  * ctor kwargs and defaults :51-53, dx = 2/(nx-1), dy = 2/(ny-1) :58, ``assert method in [...]`` :60;
  * explicit predictor quirk (y-advection uses the x-difference) :74-85, ADI along axis 0 twice :159,:165;
  * SOR: lexicographic in-place order (wavefront-equivalent), at most nit-1 sweeps, stop when
    max|p - pPrev| <= 5e-6 :183-200; ``_get_pressure`` mutates and returns the SAME ``p`` object;
  * p BCs applied once after the SOR loop :230-231; correction ignores rho :207-208;
  * ``simulate`` returns three float64 NumPy arrays [nt, nx, ny] :267-271.

Extensions (trailing kwargs, reference defaults): ``dtype`` (np.float64 = reference arithmetic,
np.float32 = fast path), ``device``.  Inputs may be NumPy arrays or device tensors, [nx, ny] or a
batch [B, nx, ny] of independent replicas (ensemble members run as one launch).
"""
import numpy as np
import torch

from . import ops
from ._util import default_device, to_dev, like_input

SOR_TOL = 5e-6      # src/chorin_fd/simulate.py:183


class NavierStokesSystem():
    def __init__(self, u_ic, v_ic, p_ic, u_bc, v_bc, p_bc,
                 nt=200, nit=50, nx=50, ny=50, dt=0.001,
                 rho=1, nu=1, beta=1.25, method='semi_implicit', dtype=np.float64, device=None,
                 advection='reference', pressure_solver='sor'):
        self.u_ic, self.v_ic, self.p_ic = u_ic, v_ic, p_ic
        self.u_bc, self.v_bc, self.p_bc = u_bc, v_bc, p_bc
        self.nt, self.nit, self.dt, self.nx, self.ny = nt, nit, dt, nx, ny
        self.dx, self.dy = 2. / (self.nx - 1), 2. / (self.ny - 1)
        self.rho, self.nu, self.beta = rho, nu, beta
        assert method in ['semi_implicit', 'explicit']
        self.method = method
        # Options beyond the reference; the defaults reproduce it bit for bit.
        #   advection='corrected'      : explicit: v d/dy along y (the reference differences along x twice);
        #                                semi_implicit: second ADI solve along axis 1 (the reference solves along axis 0 twice)
        #   pressure_solver='redblack' : red-black SOR (parallel half-sweeps) instead of the lexicographic order
        assert advection in ['reference', 'corrected'] and pressure_solver in ['sor', 'redblack']
        # (for method='semi_implicit', advection='corrected' selects the true y-direction second ADI solve)
        self.advection, self.pressure_solver = advection, pressure_solver
        self.dtype = np.dtype(dtype)
        self.device = device if device is not None else default_device()
        self._u_bcl = ops.make_bc_list(u_bc) if u_bc is not None else None
        self._v_bcl = ops.make_bc_list(v_bc) if v_bc is not None else None
        self._p_bcl = ops.make_bc_list(p_bc) if p_bc is not None else None
        self.last_sor_info = None      # device tensor [B, 2]: (sweeps, err) of the last pressure solve

    # ------------------------------------------------------------------ device-level operators
    def _d(self, x):
        return to_dev(x, self.dtype, self.device)

    def _predict_dev(self, u, v, u1, v1):
        if self.method == 'explicit':
            if self.advection == 'corrected':
                return ops.fd_predictor_explicit_corrected(u, v, u1, v1, self.dt, self.dx, self.dy, self.nu)
            return ops.fd_predictor_explicit(u, v, u1, v1, self.dt, self.dx, self.dy, self.nu)
        elif self.method == 'semi_implicit':
            return ops.fd_predictor_adi(u, v, u1, v1, self.dt, self.dx, self.dy, self.nu, corrected=self.advection == 'corrected')
        raise Exception('method not recognized: {}'.format(self.method))

    def _pressure_dev_(self, ui, vi, p):
        C = ops.fd_pressure_rhs(ui, vi, self.dt, self.dx, self.dy, self.rho)
        solve = ops.fd_sor_redblack_ if self.pressure_solver == 'redblack' else ops.fd_sor_
        kw = {} if self.pressure_solver == 'redblack' else dict(hint=self._hint_for(p))       # the previous solve's sweep count sizes the first batch
        self.last_sor_info = solve(p, C, self.dx, self.dy, self.beta, SOR_TOL, max(int(self.nit) - 1, 0), **kw)
        return p

    def _hint_for(self, p):
        """The info of the previous pressure solve when it belongs to grids of this shape (a time loop): a scheduling hint for the next solve."""
        h = self.last_sor_info
        B = 1 if p.dim() == 2 else p.shape[0]
        return h if (h is not None and h.dtype == p.dtype and h.device == p.device and tuple(h.shape) == (B, 2)) else None

    def _fused_step_applies(self, p):
        """The explicit method with the lexicographic solve on a grid whose p and right-hand side fit one workgroup's LDS (the reference's 51 x 51,
        BASELINE config 1's 64 x 64): the whole step is ONE launch (nns_fd_step_explicit_*), bitwise the separate operators.  fused_step=False on the
        instance keeps the separate launches (tests compare the two)."""
        return (getattr(self, 'fused_step', True) and self.method == 'explicit' and self.pressure_solver == 'sor'
                and None not in (self._u_bcl, self._v_bcl, self._p_bcl) and ops.fd_step_explicit_fits(p.shape[-2], p.shape[-1], p.dtype))

    def _step_dev(self, un, vn, un1, vn1, p, out=None, p_copy=None):
        """One step on device tensors; p is updated in place.  out = (u, v): fields to write the new velocities to (not inputs); p_copy: a field
        that also receives the new p (a trajectory slot)."""
        if self._fused_step_applies(p):
            u, v, self.last_sor_info = ops.fd_step_explicit(un, vn, un1, vn1, p, self._u_bcl, self._v_bcl, self._p_bcl, self.dt, self.dx, self.dy, self.rho,
                                                            self.nu, self.beta, SOR_TOL, max(int(self.nit) - 1, 0), corrected=self.advection == 'corrected',
                                                            out=out, p_copy=p_copy, hint=self._hint_for(p))
            return u, v, p
        u, v, p = self._step_dev_separate(un, vn, un1, vn1, p)
        if out is not None:
            out[0].copy_(u), out[1].copy_(v)
            u, v = out
        if p_copy is not None:
            p_copy.copy_(p)
        return u, v, p

    def _step_dev_separate(self, un, vn, un1, vn1, p):
        ui, vi = self._predict_dev(un, vn, un1, vn1)
        ops.bc_apply_(ui, self._u_bcl)
        ops.bc_apply_(vi, self._v_bcl)
        self._pressure_dev_(ui, vi, p)
        ops.bc_apply_(p, self._p_bcl)
        u, v = ops.fd_correction(ui, vi, p, self.dt, self.dx, self.dy)
        return u, v, p

    # ------------------------------------------------------------------ reference call surface
    def _explicit_predictor_step(self, u, v, u1, v1):
        ui, vi = ops.fd_predictor_explicit(self._d(u), self._d(v), self._d(u1), self._d(v1),
                                           self.dt, self.dx, self.dy, self.nu)
        return like_input(ui, u), like_input(vi, u)

    def _semi_implicit_predictor_step(self, u, v, u1, v1):
        ui, vi = ops.fd_predictor_adi(self._d(u), self._d(v), self._d(u1), self._d(v1),
                                      self.dt, self.dx, self.dy, self.nu)
        return like_input(ui, u), like_input(vi, u)

    def _get_pressure(self, ui, vi, p):
        """Mutates ``p`` in place and returns the same object (src/chorin_fd/simulate.py:193,:202)."""
        pd = self._d(p)        # shares memory with p when p already is a device tensor of the working dtype
        self._pressure_dev_(self._d(ui), self._d(vi), pd)
        if isinstance(p, torch.Tensor):
            if pd.data_ptr() != p.data_ptr():
                p.copy_(pd)
        else:
            p[...] = pd.cpu().numpy()
        return p

    def _correction_step(self, ui, vi, p):
        u, v = ops.fd_correction(self._d(ui), self._d(vi), self._d(p), self.dt, self.dx, self.dy)
        return like_input(u, ui), like_input(v, ui)

    def sor_info(self):
        """(sweeps, last err) of the most recent pressure solve (host sync), per grid."""
        info = self.last_sor_info.cpu().numpy()
        return [(int(r[0]), float(r[1])) for r in info]

    def step(self, un, vn, un1, vn1, p):
        if self.method not in ('explicit', 'semi_implicit'):
            raise Exception('method not recognized: {}'.format(self.method))
        pd = self._d(p)        # a device tensor of the working dtype is used (and mutated) in place, as the reference
        u, v, pd = self._step_dev(self._d(un), self._d(vn), self._d(un1), self._d(vn1), pd)
        if not isinstance(p, torch.Tensor):
            p[...] = pd.cpu().numpy()              # reference mutates p in _get_pressure and returns it
            return like_input(u, un), like_input(v, un), p
        if pd.data_ptr() != p.data_ptr():
            p.copy_(pd)
        return like_input(u, un), like_input(v, un), p

    def _init_variables(self):
        u, v, p = self._d(self.u_ic).clone(), self._d(self.v_ic).clone(), self._d(self.p_ic).clone()
        ops.bc_apply_(u, self._u_bcl)
        ops.bc_apply_(v, self._v_bcl)
        ops.bc_apply_(p, self._p_bcl)
        return u, v, p

    def simulate_device(self, use_graph=None):
        """The time loop with everything resident on the GPU; returns device tensors
        [nt, (B,) nx, ny] in the working dtype (no host transfer).  ``use_graph=True`` captures one step in a hipGraph
        (static state buffers) and replays it; results are bitwise identical.  It is OFF by default: measured on
        MI355X at 64^2, nit = 50 the step is bound by the SOR kernel (~600 barrier-separated pipeline steps, 0.25 ms),
        not by launch overhead -- replay 0.287 ms/step vs eager 0.270 ms/step."""
        u, v, p = self._init_variables()
        u1, v1 = u.clone(), v.clone()
        us = torch.empty((self.nt,) + tuple(u.shape), dtype=u.dtype, device=u.device)
        vs, ps = torch.empty_like(us), torch.empty_like(us)
        if use_graph:
            self._step_dev(u.clone(), v.clone(), u1.clone(), v1.clone(), p.clone())        # warm-up outside capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                _u, _v, _ = self._step_dev(u, v, u1, v1, p)                               # p is updated in place
                u1.copy_(u), v1.copy_(v)
                u.copy_(_u), v.copy_(_v)
            for n in range(self.nt):
                g.replay()
                us[n].copy_(u), vs[n].copy_(v), ps[n].copy_(p)
            return us, vs, ps
        for n in range(self.nt):
            _u, _v, p = self._step_dev(u, v, u1, v1, p, out=(us[n], vs[n]), p_copy=ps[n])     # straight into the trajectory: the next step reads it there
            u1, v1 = u, v
            u, v = _u, _v
        return us, vs, ps

    def simulate(self):
        us, vs, ps = self.simulate_device()
        f = lambda t: t.cpu().numpy().astype(np.float64, copy=False)
        return f(us), f(vs), f(ps)
