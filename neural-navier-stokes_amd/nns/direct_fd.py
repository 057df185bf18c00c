"""Direct finite-difference Navier-Stokes (cavity): mirror of the reference's
``src/direct_fd/simulate.py`` ``NavierStokesSystem`` on HIP kernels (csrc/fd_kernels.hip).

Reference semantics kept (file:line in src/direct_fd/simulate.py): ctor :46-54 (nu = 0.1 default,
no beta/method); axis 1 is x :60; exactly ``nit`` Jacobi sweeps with the p BC list after every
sweep :76-86; ``step`` mutates ``u, v, p`` IN PLACE and returns them :98,:109,:127 (so ``simulate``
mutates the caller's initial conditions :132); ``simulate`` returns float64 [nt, nx, ny] arrays.
Extensions: ``dtype``, ``device`` kwargs; NumPy or device-tensor inputs, optional batch axis.
"""
import numpy as np
import torch

from . import ops
from ._util import default_device, to_dev, like_input


class NavierStokesSystem():
    def __init__(self, u_ic, v_ic, p_ic, u_bc, v_bc, p_bc,
                 nt=200, nit=50, nx=50, ny=50, dt=0.001, rho=1, nu=0.1, dtype=np.float64, device=None):
        super().__init__()
        self.u_ic, self.v_ic, self.p_ic = u_ic, v_ic, p_ic
        self.u_bc, self.v_bc, self.p_bc = u_bc, v_bc, p_bc
        self.nt, self.dt, self.nx, self.ny = nt, dt, nx, ny
        self.dx, self.dy = 2. / (self.nx - 1), 2. / (self.ny - 1)
        self.nit, self.rho, self.nu = nit, rho, nu
        self.dtype = np.dtype(dtype)
        self.device = device if device is not None else default_device()
        self._u_bcl = ops.make_bc_list(u_bc) if u_bc is not None else None
        self._v_bcl = ops.make_bc_list(v_bc) if v_bc is not None else None
        self._p_bcl = ops.make_bc_list(p_bc) if p_bc is not None else None

    def _d(self, x):
        return to_dev(x, self.dtype, self.device)

    # ------------------------------------------------------------------ device-level
    def _step_dev_(self, u, v, p):
        """In place on device tensors u, v, p."""
        b = ops.fd_build_b(u, v, self.dt, self.dx, self.dy, self.rho)
        ops.fd_jacobi_(p, b, self.dx, self.dy, self.nit, self._p_bcl)
        un, vn = ops.fd_direct_update(u, v, p, self.dt, self.dx, self.dy, self.rho, self.nu)
        u.copy_(un)
        v.copy_(vn)
        ops.bc_apply_(u, self._u_bcl)
        ops.bc_apply_(v, self._v_bcl)
        return u, v, p

    # ------------------------------------------------------------------ reference call surface
    def _build_up_b(self, u, v):
        return like_input(ops.fd_build_b(self._d(u), self._d(v), self.dt, self.dx, self.dy, self.rho), u)

    def _pressure_poisson(self, p, b):
        """Mutates and returns ``p`` (src/direct_fd/simulate.py:78,:88)."""
        pd = self._d(p)
        own = isinstance(p, torch.Tensor) and pd.data_ptr() == p.data_ptr()
        if isinstance(p, torch.Tensor) and not own:
            pd = pd.clone()
        ops.fd_jacobi_(pd, self._d(b), self.dx, self.dy, self.nit, self._p_bcl)
        if isinstance(p, torch.Tensor):
            if not own:
                p.copy_(pd)
        else:
            p[...] = pd.cpu().numpy()
        return p

    def step(self, u, v, p):
        if isinstance(u, torch.Tensor) and all(t.is_cuda and t.is_contiguous() and t.dtype == self._d(t).dtype for t in (u, v, p)):
            return self._step_dev_(u, v, p)
        ud, vd, pd = self._d(u), self._d(v), self._d(p)
        if isinstance(u, torch.Tensor):
            ud, vd, pd = ud.clone(), vd.clone(), pd.clone()
        self._step_dev_(ud, vd, pd)
        for host, dev in ((u, ud), (v, vd), (p, pd)):          # in place, as the reference (:98,:109,:78)
            if isinstance(host, torch.Tensor):
                host.copy_(dev)
            else:
                host[...] = dev.cpu().numpy()
        return u, v, p

    def simulate_device(self, use_graph=False):
        u, v, p = self._d(self.u_ic), self._d(self.v_ic), self._d(self.p_ic)
        if isinstance(self.u_ic, torch.Tensor):
            u, v, p = u.clone(), v.clone(), p.clone()
        us = torch.empty((self.nt,) + tuple(u.shape), dtype=u.dtype, device=u.device)
        vs, ps = torch.empty_like(us), torch.empty_like(us)
        if use_graph:                                      # capture one step in a hipGraph and replay it (see chorin_fd)
            self._step_dev_(u.clone(), v.clone(), p.clone())
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_dev_(u, v, p)
            for n in range(self.nt):
                g.replay()
                us[n].copy_(u), vs[n].copy_(v), ps[n].copy_(p)
            return us, vs, ps, (u, v, p)
        for n in range(self.nt):
            self._step_dev_(u, v, p)
            us[n].copy_(u), vs[n].copy_(v), ps[n].copy_(p)
        return us, vs, ps, (u, v, p)

    def simulate(self):
        us, vs, ps, (u, v, p) = self.simulate_device()
        # the reference's simulate mutates the caller's IC arrays (no copy at :132): mirror that for NumPy ICs
        for host, dev in ((self.u_ic, u), (self.v_ic, v), (self.p_ic, p)):
            if isinstance(host, np.ndarray):
                host[...] = dev.cpu().numpy()
            elif isinstance(host, torch.Tensor):
                host.copy_(dev)
        f = lambda t: t.cpu().numpy().astype(np.float64, copy=False)
        return f(us), f(vs), f(ps)
