"""Chorin projection with Chebyshev Gauss-Lobatto collocation: mirror of the reference's
``src/chorin_spectral/simulate.py`` ``NavierStokesSystem`` (same ctor -- note: NO ``p_bc`` -- same
attribute names for the operator matrices, ``step``, ``simulate``, ``_predictor_step``,
``_correction_step`` and the matrix helpers).

One-time setup (matrix construction, ``eig`` / ``inv`` of (N-2)^2 matrices) stays on the host in NumPy /
LAPACK exactly as in the reference (:59-199); the per-step work -- ~30 dense float64 matmuls and the
elementwise assemblies -- runs on the GPU (csrc/cheb_kernels.hip: f64 MFMA GEMM + fused kernels).

The reference's matrices are kept AS THEY ARE (``D @ D.T`` at :493, ``bar_c`` called with N at :470-471,
sin arguments with N vs nodes with N-1): the reference is the spec, even where it is unsound (its
trajectories diverge) -- use at operator level.  Differences, on purpose: the process-
global ``warnings.filterwarnings('error')`` (:1-3) is not replicated; complex eigenvalues (N >= 52) raise
FloatingPointError here instead of a ComplexWarning-turned-error.
"""
import numpy as np
import torch

from . import ops
from ._util import default_device


def dup_vector_by_row(v, n):
    return v[:, np.newaxis].repeat(n, axis=1)


def dup_vector_by_col(v, n):
    return dup_vector_by_row(v, n).T


class NavierStokesSystem():
    def __init__(self, u_ic, v_ic, p_ic, u_bc, v_bc, nt=200, nit=50,
                 nx=50, ny=50, dt=0.001, rho=1, nu=1, beta=1.25, device=None, matrices='reference'):
        self.u_ic, self.v_ic, self.p_ic = u_ic, v_ic, p_ic
        self.u_bc, self.v_bc = u_bc, v_bc                      # no BC needed for pressure (:44)
        self.nt, self.nit, self.dt, self.nx, self.ny = nt, nit, dt, nx, ny
        self.dx, self.dy = 2. / self.nx, 2. / self.ny          # (:48)
        self.rho, self.nu, self.beta = rho, nu, beta
        self.device = device if device is not None else default_device()
        # matrices='corrected' (an option of the build; the default reproduces the
        # reference): D and T^-1 built for the polynomial degree N - 1 that the N Gauss-Lobatto nodes of :398 carry
        # (the reference uses N at :436-440 and :470-472, which differentiates no polynomial exactly), and D^2 = D @ D
        # (the FIXME at :493).  The rest of the scheme -- boundary folding, nu-free predictor, pressure update -- is the
        # reference's.
        assert matrices in ['reference', 'corrected']
        self.matrices = matrices
        self._pseudospectral_setup()

    # ------------------------------------------------------------------ matrix helpers (:387-531)
    def _get_c_k(self, k):
        assert k >= 0
        return 2 if k == 0 else 1

    def _get_bar_c_k(self, k, N):
        assert k >= 0
        return 2 if (k == 0 or k == N) else 1

    def _get_gauss_lobatto_points(self, N, k=1):
        return np.cos(k * np.pi * np.arange(N) / float(N - 1))

    def _get_T_matrix(self, N):
        return np.stack([self._get_gauss_lobatto_points(N, k=k) for k in np.arange(0, N)])

    def _degree(self, N):
        return N - 1 if self.matrices == 'corrected' else N

    def _get_inv_T_matrix(self, N):
        M = self._degree(N)
        inv_T = self._get_T_matrix(N).T
        bar_c_i = np.stack([np.repeat(self._get_bar_c_k(i, M), N) for i in np.arange(0, N)])
        return 2 * inv_T / (bar_c_i.T * bar_c_i * M)

    def _get_D_matrix(self, N):
        idx = np.arange(N)
        i, j = idx[:, None].astype(np.float64), idx[None, :].astype(np.float64)
        M = self._degree(N)
        bc = np.array([self._get_bar_c_k(k, M) for k in range(N)], dtype=np.float64)
        sign = np.where((idx[:, None] + idx[None, :]) % 2 == 0, 1.0, -1.0)
        with np.errstate(divide='ignore', invalid='ignore'):
            diff = 2 * np.sin((j + i) * np.pi / (2. * M)) * np.sin((j - i) * np.pi / (2. * M))
            D = bc[:, None] / bc[None, :] * sign / diff
        D[idx, idx] = 0.0
        for r in range(N):
            D[r, r] = -np.sum(D[r, :])
        return D

    def _get_D_sqr_matrix(self, N):
        D = self._get_D_matrix(N)
        if self.matrices == 'corrected':
            return D @ D
        D_sqr = (D @ D.T).copy()                               # FIXME in the reference (:493); kept
        for r in range(N):
            D_sqr[r, r] = -np.sum(D_sqr[r, :])                 # row sum still contains the old diagonal (:502)
        return D_sqr

    def _get_D_matrix_degrees_minus_2(self, N):
        """The reference's interior pressure-derivative matrix (src/chorin_spectral/simulate.py:506-531): for interior nodes i != j
        D_ij = (-1)^(j+1) (1 - x_j^2) / ((1 - x_i^2)(x_i - x_j)), D_ii = 3 x_i / (2 (1 - x_i^2)); as array expressions."""
        x = self._get_gauss_lobatto_points(N)[1:-1]
        w = 1. - x ** 2
        sign = np.where(np.arange(1, N - 1) % 2 == 0, -1.0, 1.0)               # (-1)^(j+1) at the global node index j
        with np.errstate(divide='ignore', invalid='ignore'):
            D = (sign * w)[None, :] / (w[:, None] * (x[:, None] - x[None, :]))
        np.fill_diagonal(D, 3 * x / (2. * w))
        return D

    def _get_D_matrix_interior_lagrange(self, N):
        """matrices='corrected' only: the exact derivative matrix of the interpolant on the N-2 interior nodes (barycentric
        weights), in place of _get_D_matrix_degrees_minus_2, whose rows do not even annihilate constants (row sums ~10)."""
        x = self._get_gauss_lobatto_points(N)[1:-1]
        n = N - 2
        w = np.array([1. / np.prod([x[j] - x[m] for m in range(n) if m != j]) for j in range(n)])
        D = np.zeros((n, n))
        for i in range(n):
            for j in range(n):
                if i != j:
                    D[i, j] = w[j] / w[i] / (x[i] - x[j])
            D[i, i] = -np.sum(D[i, :])
        return D

    def _process_boundary_conditions(self, bc_list):
        """alpha u + beta du/dx = g per side (:201-230).  Neumann data (alpha = 0, beta = 1, g = the derivative along the +axis
        direction, as src/boundary.py:56-86 defines it) is accepted with matrices='corrected' only; the default raises
        NotImplementedError like the reference (:218-221)."""
        vals = {}
        names = {'left': 'minus_x', 'right': 'plus_x', 'top': 'minus_y', 'bottom': 'plus_y'}     # (:204-215)
        for s in names.values():
            vals['beta_' + s] = 0
        for bc in bc_list:
            if bc.type not in ('dirichlet', 'neumann'):
                raise Exception('Boundary type {} not supported'.format(bc.type))
            if bc.type == 'neumann' and self.matrices != 'corrected':
                raise NotImplementedError                       # (:218-221)
            if bc.boundary not in names:
                raise Exception('Boundary side {} not supported'.format(bc.boundary))
            neu = bc.type == 'neumann'
            vals['alpha_' + names[bc.boundary]], vals['beta_' + names[bc.boundary]] = (0, 1) if neu else (1, 0)
            vals['g_' + names[bc.boundary]] = bc.value
        return vals

    @staticmethod
    def _fold(A, k):
        """(A u)_int = A_fold u_int + const with the two boundary values eliminated (oracle.chorin_spectral.Setup.fold)."""
        Af = A[1:-1, 1:-1] + 1. / k['e'] * (np.outer(A[1:-1, 0], k['b0']) + np.outer(A[1:-1, -1], k['bN']))
        return Af, A[1:-1, 0] * k['c0'] + A[1:-1, -1] * k['cN']

    @staticmethod
    def _boundary_constants(D, bc, ax):
        am, ap = bc['alpha_minus_' + ax], bc['alpha_plus_' + ax]
        bm, bp = bc['beta_minus_' + ax], bc['beta_plus_' + ax]
        c0_minus = -bp * D[0, -1]
        c0_plus = am + bm * D[-1, -1]
        cN_plus = -bm * D[-1, 0]
        cN_minus = ap + bp * D[0, 0]
        e = c0_plus * cN_minus - c0_minus * cN_plus
        b0 = -c0_plus * bp * D[0, 1:-1] - c0_minus * bm * D[-1, 1:-1]
        bN = -cN_minus * bm * D[-1, 1:-1] - cN_plus * bp * D[0, 1:-1]
        return dict(e=e, c0_minus=c0_minus, c0_plus=c0_plus, cN_minus=cN_minus, cN_plus=cN_plus, b0=b0, bN=bN)

    @staticmethod
    def _real_eig(M, what):
        lam, P = np.linalg.eig(M)
        if np.iscomplexobj(lam) and np.abs(lam.imag).max() > 0:
            raise FloatingPointError("chorin_spectral: complex eigenvalues in %s (the reference fails here too, for N >= 52)" % what)
        return np.real(lam), np.real(P)

    def _dev(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=self.device)

    def _pseudospectral_setup(self):
        Nx, Ny = self.nx, self.ny
        self.x_i, self.y_i = self._get_gauss_lobatto_points(Nx), self._get_gauss_lobatto_points(Ny)
        self.Tx, self.Ty = self._get_T_matrix(Nx), self._get_T_matrix(Ny)
        self.Tx_inv, self.Ty_inv = self._get_inv_T_matrix(Nx), self._get_inv_T_matrix(Ny)
        self.Dx, self.Dy = self._get_D_matrix(Nx), self._get_D_matrix(Ny)
        self.Dx_sqr, self.Dy_sqr = self._get_D_sqr_matrix(Nx), self._get_D_sqr_matrix(Ny)
        self._bc = {'u': self._process_boundary_conditions(self.u_bc), 'v': self._process_boundary_conditions(self.v_bc)}
        self._k, self._helm = {}, {}
        for f in ('u', 'v'):
            kx = self._boundary_constants(self.Dx, self._bc[f], 'x')
            ky = self._boundary_constants(self.Dy, self._bc[f], 'y')
            g = self._bc[f]
            for k, ax in ((kx, 'x'), (ky, 'y')):               # the g-dependent constants of the two boundary values
                k['c0'] = (k['c0_minus'] * g['g_minus_' + ax] + k['c0_plus'] * g['g_plus_' + ax]) / k['e']
                k['cN'] = (k['cN_minus'] * g['g_minus_' + ax] + k['cN_plus'] * g['g_plus_' + ax]) / k['e']
            self._k[f] = (kx, ky)
            if self.matrices == 'corrected':
                # every derivative sees the boundary values: operators with the boundary rows folded in (outer products) and
                # the constants the inhomogeneous data leave behind, as [n, n] device matrices the GEMMs accumulate onto
                Mx, kxx = self._fold(self.Dx_sqr, kx)
                My, kyy = self._fold(self.Dy_sqr, ky)
                D1x, k1x = self._fold(self.Dx, kx)
                D1y, k1y = self._fold(self.Dy, ky)
                ones_r, ones_c = np.ones((1, Ny - 2)), np.ones((Nx - 2, 1))
                self._fold_dev = getattr(self, '_fold_dev', {})
                self._fold_dev[f] = dict(Dx=self._dev(D1x), Dy=self._dev(D1y), Mx=self._dev(Mx), My=self._dev(My),
                                         cx=self._dev(k1x[:, None] * ones_r), cy=self._dev(ones_c * k1y[None, :]),
                                         cxx2=self._dev(2 * kxx[:, None] * ones_r), cyy2=self._dev(2 * ones_c * kyy[None, :]))
            else:
                Mx = self.Dx_sqr[1:-1, 1:-1] + 1. / kx['e'] * (kx['b0'] * self.Dx_sqr[1:-1, 0] + kx['bN'] * self.Dx_sqr[1:-1, -1])
                My = self.Dy_sqr[1:-1, 1:-1] + 1. / ky['e'] * (ky['b0'] * self.Dy_sqr[1:-1, 0] + ky['bN'] * self.Dy_sqr[1:-1, -1])
            lx, P = self._real_eig(Mx, f + '_Dx')
            ly, Q = self._real_eig(My, f + '_Dy')
            for name, val in (('Dx_lambda', lx), ('Dx_P', P), ('Dy_lambda', ly), ('Dy_Q', Q),
                              ('Dx_P_inv', np.linalg.inv(P)), ('Dy_Q_inv', np.linalg.inv(Q))):
                setattr(self, '%s_%s' % (f, name), val)                     # reference attribute names (:174-183)
            self._helm[f] = {k: self._dev(v) for k, v in dict(lx=lx, ly=ly, P=P, Q=Q, P_inv=np.linalg.inv(P), Q_inv=np.linalg.inv(Q)).items()}
        DP = self._get_D_matrix_interior_lagrange if self.matrices == 'corrected' else self._get_D_matrix_degrees_minus_2
        self.DPx, self.DPy = DP(Nx), DP(Ny)
        self.DxDPx = self.Dx[1:-1, 1:-1] @ self.DPx
        self.DyDPy = self.Dy[1:-1, 1:-1] @ self.DPy
        self.DxDPx_lambda, self.DxDPx_P = self._real_eig(self.DxDPx, 'DxDPx')
        self.DyDPy_lambda, self.DyDPy_Q = self._real_eig(self.DyDPy, 'DyDPy')
        self.DxDPx_P_inv, self.DyDPy_Q_inv = np.linalg.inv(self.DxDPx_P), np.linalg.inv(self.DyDPy_Q)
        d = self._dev
        self._D = dict(Dx=d(self.Dx[1:-1, 1:-1]), Dy=d(self.Dy[1:-1, 1:-1]), Dxx=d(self.Dx_sqr[1:-1, 1:-1]), Dyy=d(self.Dy_sqr[1:-1, 1:-1]),
                       DxDPx=d(self.DxDPx), DyDPy=d(self.DyDPy), lpx=d(self.DxDPx_lambda), lpy=d(self.DyDPy_lambda),
                       PP=d(self.DxDPx_P), PQ=d(self.DyDPy_Q), PP_inv=d(self.DxDPx_P_inv), PQ_inv=d(self.DyDPy_Q_inv))
        # the correction step's boundary source S (:353-361) depends on the BC values only
        gu, gv = self._bc['u'], self._bc['v']
        u_tau = np.stack([np.ones(Ny - 2) * gu['g_minus_x'], np.ones(Ny - 2) * gu['g_plus_x']])
        v_tau = np.stack([np.ones(Nx - 2) * gv['g_minus_y'], np.ones(Nx - 2) * gv['g_plus_y']]).T
        Dx_bar = np.stack([self.Dx[1:-1, 0], self.Dx[1:-1, -1]]).T
        Dy_bar = np.stack([self.Dy[1:-1, 0], self.Dy[1:-1, -1]]).T
        self._S = d(-(Dx_bar @ u_tau + v_tau @ Dy_bar.T))
        if self.matrices == 'corrected':
            lam = self.DxDPx_lambda[:, None] + self.DyDPy_lambda[None, :]
            self._null = np.argwhere(np.abs(lam) <= 1e-9 * np.abs(lam).max())         # the constant pressure: lambda_x0 + lambda_y0 = 0
            self._Dfull = dict(Dx=d(self.Dx[1:-1, :]), Dy=d(self.Dy[1:-1, :]), DPx=d(self.DPx), DPy=d(self.DPy))

    # ------------------------------------------------------------------ per-step operators on the GPU
    def _boundary_vectors(self, sol, name):
        """get_boundary_values (:245-256): with Dirichlet-only BCs b0 = bN = 0 and these are constants; the general
        b0 / bN terms are kept as (1 x n) @ (n x n) GEMMs on the device."""
        kx, ky = self._k[name]
        g = self._bc[name]
        ni, nj = sol.shape
        cache = self.__dict__.setdefault('_bvec_dev', {})       # the b-vectors are constants of the set-up: on the device ONCE (they were copied from
                                                                # the host in every step, which also kept a step from being captured as a HIP graph)
        def vec(key, b, const, n, left):
            out = torch.full((n,), float(const), dtype=torch.float64, device=self.device)
            if np.any(b != 0):
                bt = cache.get((name, key))
                if bt is None:
                    bt = cache[(name, key)] = self._dev(b[None, :] if left else b[:, None])
                prod = ops.cheb_gemm(bt, sol) if left else ops.cheb_gemm(sol, bt)
                out = out + prod.reshape(-1)
            return out.contiguous()
        corr = self.matrices == 'corrected'       # the reference leaves the constant of the last row / column out (:250,:253)
        x0 = vec('x0', kx['b0'] / kx['e'], (kx['c0_minus'] * g['g_minus_x'] + kx['c0_plus'] * g['g_plus_x']) / kx['e'], nj, True)
        xN = vec('xN', kx['bN'] / kx['e'], kx['cN'] if corr else 0.0, nj, True)
        y0 = vec('y0', ky['b0'] / ky['e'], (ky['c0_minus'] * g['g_minus_y'] + ky['c0_plus'] * g['g_plus_y']) / ky['e'], ni, False)
        yN = vec('yN', ky['bN'] / ky['e'], ky['cN'] if corr else 0.0, ni, False)
        return x0, xN, y0, yN

    def _predict_dev(self, un, vn, un1, vn1):
        D, mm = self._D, ops.cheb_gemm
        I = lambda a: a[1:-1, 1:-1].contiguous()
        _un, _vn, _un1, _vn1 = I(un), I(vn), I(un1), I(vn1)
        out = []
        for name, f, f1 in (('u', _un, _un1), ('v', _vn, _vn1)):
            if self.matrices == 'corrected':
                # folded operators + the constants of inhomogeneous data; cxx2 / cyy2 carry the explicit AND the implicit
                # side's share (oracle.chorin_spectral.predictor_step); all-zero constants for homogeneous Dirichlet data
                o = self._fold_dev[name]
                acc = lambda c, A, B, tb: mm(A, B, transB=tb, beta=1.0, out=c.clone())
                fx, fy = acc(o['cx'], o['Dx'], f, False), acc(o['cy'], f, o['Dy'], True)
                f1x, f1y = acc(o['cx'], o['Dx'], f1, False), acc(o['cy'], f1, o['Dy'], True)
                fxx, fyy = acc(o['cxx2'], o['Mx'], f, False), acc(o['cyy2'], f, o['My'], True)
            else:
                fx, fy = mm(D['Dx'], f), mm(f, D['Dy'], transB=True)                       # (:264-268)
                f1x, f1y = mm(D['Dx'], f1), mm(f1, D['Dy'], transB=True)
                fxx, fyy = mm(D['Dxx'], f), mm(f, D['Dyy'], transB=True)                   # (:270-274)
            F = ops.cheb_helmholtz_rhs(f, _un, _vn, _un1, _vn1, fx, fy, f1x, f1y, fxx, fyy, self.dt)
            h = self._helm[name]
            Hh = mm(mm(h['P_inv'], F), h['Q_inv'], transB=True)                        # (:285-286)
            hat = ops.cheb_diag_div(Hh, h['lx'], h['ly'], 2.0, -self.dt, -self.dt)     # (:287-288)
            sol = mm(h['P'], mm(hat, h['Q'], transB=True))                             # (:289-290)
            out.append(ops.cheb_embed(sol, *self._boundary_vectors(sol, name)))
        return out[0], out[1]

    def _correct_dev_corrected(self, ui, vi, p):
        """matrices='corrected': the projection the reference's :339-383 is after (oracle.chorin_spectral.correction_step_corrected):
        interior divergence from the boundary values u* carries (full rows of D), the constant pressure mode projected out of
        the Uzawa solve, and the velocity update by the pressure GRADIENT (dt / rho) DP Q (the reference subtracts DxDPx Q)."""
        D, F, mm = self._D, self._Dfull, ops.cheb_gemm
        div = mm(F['Dx'], ui[:, 1:-1].contiguous())                                                # Dx[1:-1, :] @ u*[:, 1:-1]
        mm(vi[1:-1, :].contiguous(), F['Dy'], transB=True, beta=1.0, out=div)                      # + v*[1:-1, :] @ Dy[1:-1, :]^T
        Hh = mm(mm(D['PP_inv'], div, alpha=self.rho / self.dt), D['PQ_inv'], transB=True)
        Qh = ops.cheb_diag_div(Hh, D['lpx'], D['lpy'], 0.0, 1.0, 1.0)
        if len(self._null):                                                                         # pressure is defined up to a constant
            mask = self.__dict__.get('_null_mask_dev')
            if mask is None:                                                                        # (built once, outside any graph capture: the warm-up step)
                m = np.zeros(tuple(Qh.shape), dtype=bool)
                for i, j in self._null:
                    m[i, j] = True
                mask = self._null_mask_dev = torch.as_tensor(m, device=Qh.device)
            Qh.masked_fill_(mask, 0.0)
        Q = mm(D['PP'], mm(Qh, D['PQ'], transB=True))
        I = lambda a: a[1:-1, 1:-1].contiguous()
        ui_i, vi_i = I(ui), I(vi)
        mm(F['DPx'], Q, alpha=-self.dt / self.rho, beta=1.0, out=ui_i)
        mm(Q, F['DPy'], transB=True, alpha=-self.dt / self.rho, beta=1.0, out=vi_i)
        u1, v1, p1 = ui.clone(), vi.clone(), p.clone()
        u1[1:-1, 1:-1] = ui_i
        v1[1:-1, 1:-1] = vi_i
        p1[1:-1, 1:-1] = Q
        return u1, v1, p1

    def _correct_dev(self, ui, vi, p):
        if self.matrices == 'corrected':
            return self._correct_dev_corrected(ui, vi, p)
        D, mm = self._D, ops.cheb_gemm
        I = lambda a: a[1:-1, 1:-1].contiguous()
        ui_i, vi_i = I(ui), I(vi)
        Tm = self._S.clone()                                                           # S - Dx ui - vi Dy^T   (:367)
        mm(D['Dx'], ui_i, alpha=-1.0, beta=1.0, out=Tm)
        mm(vi_i, D['Dy'], transB=True, alpha=-1.0, beta=1.0, out=Tm)
        Ht = mm(D['PP_inv'], Tm, alpha=-self.rho / self.dt)                            # H_tilde = PP_inv @ (-rho/dt * T)
        Hh = mm(Ht, D['PQ_inv'], transB=True)
        Qh = ops.cheb_diag_div(Hh, D['lpx'], D['lpy'], 0.0, 1.0, 1.0)                  # (:372-373)
        Q = mm(D['PP'], mm(Qh, D['PQ'], transB=True))
        u1, v1, p1 = ui.clone(), vi.clone(), p.clone()
        mm(D['DxDPx'], Q, alpha=-self.dt / self.rho, beta=1.0, out=ui_i)               # (:379)
        mm(Q, D['DyDPy'], transB=True, alpha=-self.dt / self.rho, beta=1.0, out=vi_i)  # (:380)
        u1[1:-1, 1:-1] = ui_i
        v1[1:-1, 1:-1] = vi_i
        p1[1:-1, 1:-1] = Q
        return u1, v1, p1

    # ------------------------------------------------------------------ reference call surface
    def _predictor_step(self, un, vn, un1, vn1):
        ui, vi = self._predict_dev(*[self._dev(a) for a in (un, vn, un1, vn1)])
        return ui.cpu().numpy(), vi.cpu().numpy()

    def _correction_step(self, ui, vi, p):
        a, b, c = self._correct_dev(self._dev(ui), self._dev(vi), self._dev(p))
        return a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()

    def step(self, un, vn, un1, vn1, p):
        ui, vi = self._predict_dev(*[self._dev(a) for a in (un, vn, un1, vn1)])
        a, b, c = self._correct_dev(ui, vi, self._dev(p))
        return a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()

    def _init_variables(self):
        from .boundary import apply_list
        u, v, p = np.array(self.u_ic, dtype=np.float64), np.array(self.v_ic, dtype=np.float64), np.array(self.p_ic, dtype=np.float64)
        apply_list(u, self.u_bc)
        apply_list(v, self.v_bc)
        return u, v, p

    def simulate(self, use_graph=None):
        """The time loop (:343-362).  A step is ~90 launches of 2-7 us on 51 x 51 matrices -- bound by what the HOST needs to enqueue them (0.56 ms
        per step, profiles/r04_small_paths.txt) -- and every step runs the same kernels on the same buffers, so by default ONE step is captured
        as a HIP graph and replayed (use_graph=False: the eager loop; a capture that fails falls back to it).  Same kernels, same order: the
        trajectories are bitwise those of the eager loop (test)."""
        u, v, p = (self._dev(a) for a in self._init_variables())
        u1, v1 = u.clone(), v.clone()
        if use_graph is None:
            use_graph = u.is_cuda
        graph = None
        if use_graph and self.nt > 2:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                       # lazy initialisation inside the launchers happens here, not in the capture
                    a, b = self._predict_dev(u.clone(), v.clone(), u1.clone(), v1.clone())
                    self._correct_dev(a, b, p.clone())
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    ui, vi = self._predict_dev(u, v, u1, v1)
                    _u, _v, _p = self._correct_dev(ui, vi, p)
                    u1.copy_(u), v1.copy_(v)
                    u.copy_(_u), v.copy_(_v), p.copy_(_p)
            except Exception as e:                                   # noqa: BLE001 -- the eager loop is the same computation
                print('chorin_spectral: HIP graph capture of the step failed (%r): running eagerly' % (e,))
                graph = None
                u, v, p = (self._dev(a) for a in self._init_variables())
                u1, v1 = u.clone(), v.clone()
        self.last_simulate_used_graph = graph is not None
        if graph is not None:
            us = torch.empty((self.nt,) + tuple(u.shape), dtype=u.dtype, device=u.device)
            vs, ps = torch.empty_like(us), torch.empty_like(us)
            for n in range(self.nt):
                graph.replay()
                us[n].copy_(u), vs[n].copy_(v), ps[n].copy_(p)
            return us.cpu().numpy(), vs.cpu().numpy(), ps.cpu().numpy()
        us, vs, ps = [], [], []
        for n in range(self.nt):
            ui, vi = self._predict_dev(u, v, u1, v1)
            _u, _v, p = self._correct_dev(ui, vi, p)
            u1, v1 = u, v
            u, v = _u, _v
            us.append(u), vs.append(v), ps.append(p)
        f = lambda l: torch.stack(l).cpu().numpy()
        return f(us), f(vs), f(ps)
