"""Oracle (test infrastructure only): Chebyshev-collocation Chorin projection.

NumPy restatement of ``src/chorin_spectral/simulate.py`` of the reference: the dense
differentiation / transform matrices and one predictor / correction application.  The
reference's matrices are known to be unsound (``D @ D.T`` instead of ``D @ D`` at :493, the
``bar_c`` weight never fires for the last point because it is called with N instead of N-1 at
:470-471, sin arguments use N while the nodes use N-1 at :398 vs :472) and trajectories
diverge -- they are restated here *as they are*, because the reference is
the spec; use at operator level only (matrix construction at any N, single steps at N <= 51
where the eigenvalues are real).

Boundary-side naming follows :203-215: 'left' -> (minus, x), 'right' -> (plus, x),
'top' -> (minus, y), 'bottom' -> (plus, y).  Neumann raises NotImplementedError (:218-221).
"""
import numpy as np


def bar_c(k, N):
    """src/chorin_spectral/simulate.py:391-393"""
    return 2 if (k == 0 or k == N) else 1


def gauss_lobatto_points(N, k=1):
    """:395-399"""
    return np.cos(k * np.pi * np.arange(N) / float(N - 1))


def T_matrix(N):
    """:401-419  T[k, i] = cos(k pi i / (N-1))"""
    return np.stack([gauss_lobatto_points(N, k=k) for k in range(N)])


def inv_T_matrix(N, corrected=False):
    """:421-441.  corrected: the weights and the normalisation use the polynomial degree N - 1 (then inv_T @ T = I)."""
    M = N - 1 if corrected else N
    inv_T = T_matrix(N).T
    bc = np.array([bar_c(i, M) for i in range(N)], dtype=np.int64)
    bar_c_i = np.repeat(bc[:, None], N, axis=1)
    bar_c_k = bar_c_i.T
    return 2 * inv_T / (bar_c_k * bar_c_i * M)


def D_matrix(N, corrected=False):
    """:443-481 (off-diagonals from the sin form, diagonal by the negative-sum trick).
    corrected (an option of the build): the N nodes x_j = cos(pi j / (N - 1)) of :398 carry
    polynomials of degree N - 1, so the end-point weight is bar_c(k, N - 1) and x_i - x_j = 2 sin((i+j) pi / (2 (N-1)))
    sin((j-i) pi / (2 (N-1))): the reference uses N in both places (:470-472), which differentiates nothing exactly."""
    M = N - 1 if corrected else N
    i = np.arange(N)[:, None].astype(np.float64)
    j = np.arange(N)[None, :].astype(np.float64)
    bc = np.array([bar_c(k, M) for k in range(N)], dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        diff = 2 * np.sin((j + i) * np.pi / (2. * M)) * np.sin((j - i) * np.pi / (2. * M))
        sign = np.where(((np.arange(N)[:, None] + np.arange(N)[None, :]) % 2) == 0, 1.0, -1.0)
        D = bc[:, None] / bc[None, :] * sign / diff
    D[np.arange(N), np.arange(N)] = 0.0
    for r in range(N):
        D[r, r] = -np.sum(D[r, :])
    return D


def D_sqr_matrix(N, corrected=False):
    """:483-504.  NOTE the diagonal: the row sum taken at :502 still contains the D D^T
    diagonal entry, so d_ii = -sum_j (D D^T)_ij over ALL j (the in-code comment is wrong).
    corrected: the second-derivative matrix IS D @ D (the FIXME at :493)."""
    if corrected:
        D = D_matrix(N, corrected=True)
        return D @ D
    D = D_matrix(N)
    D_sqr = (D @ D.T).copy()
    for r in range(N):
        D_sqr[r, r] = -np.sum(D_sqr[r, :])
    return D_sqr


def D_matrix_degrees_minus_2(N):
    """:506-531  (N-2)x(N-2) pressure derivative matrix of the P_N - P_{N-2} method."""
    D = np.zeros((N, N))
    x = gauss_lobatto_points(N)
    for i in range(1, N - 1):
        for j in range(1, N - 1):
            if i != j:
                D[i, j] = ((-1)**(j + 1) * (1. - x[j]**2) / ((1. - x[i]**2) * (x[i] - x[j])))
            else:
                D[i, i] = 3 * x[i] / (2. * (1. - x[i]**2))
    return D[1:-1, 1:-1]


def D_matrix_interior_lagrange(N):
    """Corrected pressure derivative (an option of the build): the EXACT derivative matrix of the degree N-3 interpolant on
    the N-2 interior Gauss-Lobatto nodes (barycentric weights; negative-sum diagonal).  The reference's formula (:506-531,
    D_matrix_degrees_minus_2 above) does not differentiate constants (row sums of order 10)."""
    x = gauss_lobatto_points(N)[1:-1]
    n = N - 2
    w = np.array([1. / np.prod([x[j] - x[m] for m in range(n) if m != j]) for j in range(n)])
    D = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i != j:
                D[i, j] = w[j] / w[i] / (x[i] - x[j])
        D[i, i] = -np.sum(D[i, :])
    return D


def process_boundary_conditions(bc_list, allow_neumann=False):
    """:201-230.  bc_list = [(kind, side, value, dx, dy), ...] -> dict of alpha/beta/g  (alpha u + beta du/dx = g on each side).
    allow_neumann (the corrected path only; the reference raises NotImplementedError, :218-221): alpha = 0, beta = 1, g = the
    derivative along the +axis direction, as src/boundary.py:56-86 defines the Neumann value."""
    out = {}
    names = {'left': 'minus_x', 'right': 'plus_x', 'top': 'minus_y', 'bottom': 'plus_y'}
    for s in names.values():
        out['beta_' + s] = 0
    for (kind, side, value, _dx, _dy) in bc_list:
        if kind == 'dirichlet':
            if side not in names:
                raise Exception('Boundary side {} not supported'.format(side))
            out['alpha_' + names[side]], out['beta_' + names[side]] = 1, 0
            out['g_' + names[side]] = value
        elif kind == 'neumann':
            if not allow_neumann:
                raise NotImplementedError
            if side not in names:
                raise Exception('Boundary side {} not supported'.format(side))
            out['alpha_' + names[side]], out['beta_' + names[side]] = 0, 1
            out['g_' + names[side]] = value
        else:
            raise Exception('Boundary type {} not supported'.format(kind))
    return out


def boundary_constants(D, bc, ax):
    """get_boundary_constants :102-118 for axis ``ax`` in {'x','y'}."""
    am, ap = bc['alpha_minus_' + ax], bc['alpha_plus_' + ax]
    bm, bp = bc['beta_minus_' + ax], bc['beta_plus_' + ax]
    c0_minus = -bp * D[0, -1]
    c0_plus = am + bm * D[-1, -1]
    cN_plus = -bm * D[-1, 0]
    cN_minus = ap + bp * D[0, 0]
    e = c0_plus * cN_minus - c0_minus * cN_plus
    b0 = -c0_plus * bp * D[0, 1:-1] - c0_minus * bm * D[-1, 1:-1]
    bN = -cN_minus * bm * D[-1, 1:-1] - cN_plus * bp * D[0, 1:-1]
    return dict(e=e, c0_minus=c0_minus, c0_plus=c0_plus, cN_minus=cN_minus, cN_plus=cN_plus,
                b0=b0, bN=bN)


class Setup(object):
    """_pseudospectral_setup :59-199 as a plain container of arrays."""

    def __init__(self, nx, ny, u_bc, v_bc, corrected=False):
        """corrected=True: D, D^2 and T^-1 from the corrected constructors (the rest of the scheme is the reference's)."""
        self.nx, self.ny = nx, ny
        self.corrected = corrected
        if corrected:
            self._build(nx, ny, u_bc, v_bc, lambda N: inv_T_matrix(N, True), lambda N: D_matrix(N, True), lambda N: D_sqr_matrix(N, True))
        else:
            self._build(nx, ny, u_bc, v_bc, inv_T_matrix, D_matrix, D_sqr_matrix)

    @staticmethod
    def fold(A, k):
        """Boundary values eliminated from the operator A (N x N) acting along one axis: with u_0 = c0 + (b0 . u_int) / e and
        u_N = cN + (bN . u_int) / e, (A u)_int = A_fold u_int + const;  returns (A_fold [(N-2)^2], const [N-2]) given the g-dependent
        constants c0, cN in k.  (The reference adds the VECTORS b0 * A[1:-1, 0] to the matrix rows, :159-166 -- an outer product
        is meant; both vanish for Dirichlet sides.)"""
        Af = A[1:-1, 1:-1] + 1. / k['e'] * (np.outer(A[1:-1, 0], k['b0']) + np.outer(A[1:-1, -1], k['bN']))
        return Af, A[1:-1, 0] * k['c0'] + A[1:-1, -1] * k['cN']

    def _build(self, nx, ny, u_bc, v_bc, inv_T_matrix, D_matrix, D_sqr_matrix):
        self.x_i, self.y_i = gauss_lobatto_points(nx), gauss_lobatto_points(ny)
        self.Tx, self.Ty = T_matrix(nx), T_matrix(ny)
        self.Tx_inv, self.Ty_inv = inv_T_matrix(nx), inv_T_matrix(ny)
        self.Dx, self.Dy = D_matrix(nx), D_matrix(ny)
        self.Dx_sqr, self.Dy_sqr = D_sqr_matrix(nx), D_sqr_matrix(ny)
        c = self.corrected
        self.bc = {'u': process_boundary_conditions(u_bc, allow_neumann=c), 'v': process_boundary_conditions(v_bc, allow_neumann=c)}
        self.k = {}
        self.helm = {}
        self.folded = {}
        for f in ('u', 'v'):
            kx = boundary_constants(self.Dx, self.bc[f], 'x')
            ky = boundary_constants(self.Dy, self.bc[f], 'y')
            g = self.bc[f]
            for k, ax in ((kx, 'x'), (ky, 'y')):               # the g-dependent constants of the two boundary values
                k['c0'] = (k['c0_minus'] * g['g_minus_' + ax] + k['c0_plus'] * g['g_plus_' + ax]) / k['e']
                k['cN'] = (k['cN_minus'] * g['g_minus_' + ax] + k['cN_plus'] * g['g_plus_' + ax]) / k['e']
            self.k[f] = (kx, ky)
            if c:
                Mx, kxx = self.fold(self.Dx_sqr, kx)
                My, kyy = self.fold(self.Dy_sqr, ky)
                D1x, k1x = self.fold(self.Dx, kx)
                D1y, k1y = self.fold(self.Dy, ky)
                self.folded[f] = dict(Dx=D1x, Dy=D1y, cx=k1x, cy=k1y, cxx=kxx, cyy=kyy)
            else:
                Mx = self.Dx_sqr[1:-1, 1:-1] + 1. / kx['e'] * (kx['b0'] * self.Dx_sqr[1:-1, 0] +
                                                              kx['bN'] * self.Dx_sqr[1:-1, -1])
                My = self.Dy_sqr[1:-1, 1:-1] + 1. / ky['e'] * (ky['b0'] * self.Dy_sqr[1:-1, 0] +
                                                              ky['bN'] * self.Dy_sqr[1:-1, -1])
            lx, P = np.linalg.eig(Mx)                  # :174-177
            ly, Q = np.linalg.eig(My)
            self.helm[f] = dict(Mx=Mx, My=My, lx=lx, P=P, P_inv=np.linalg.inv(P),
                                ly=ly, Q=Q, Q_inv=np.linalg.inv(Q))
        DP = D_matrix_interior_lagrange if c else D_matrix_degrees_minus_2
        self.DPx = DP(nx)         # :190-199
        self.DPy = DP(ny)
        self.DxDPx = self.Dx[1:-1, 1:-1] @ self.DPx
        self.DyDPy = self.Dy[1:-1, 1:-1] @ self.DPy
        self.lpx, self.PP = np.linalg.eig(self.DxDPx)
        self.lpy, self.PQ = np.linalg.eig(self.DyDPy)
        self.PP_inv, self.PQ_inv = np.linalg.inv(self.PP), np.linalg.inv(self.PQ)


def _boundary_values(sol, g, kx, ky, corrected=False):
    """get_boundary_values :245-256.  corrected: the last row / column also gets its constant (cN- g- + cN+ g+) / e, which
    the reference leaves out (its xN, yN are 0 for Dirichlet data)."""
    x0 = 1. / kx['e'] * np.sum(kx['b0'][:, None] * sol, axis=0) + \
        1. / kx['e'] * (kx['c0_minus'] * g['g_minus_x'] + kx['c0_plus'] * g['g_plus_x'])
    xN = 1. / kx['e'] * np.sum(kx['bN'][:, None] * sol, axis=0)
    y0 = 1. / ky['e'] * np.sum(ky['b0'][None, :] * sol, axis=1) + \
        1. / ky['e'] * (ky['c0_minus'] * g['g_minus_y'] + ky['c0_plus'] * g['g_plus_y'])
    yN = 1. / ky['e'] * np.sum(ky['bN'][None, :] * sol, axis=1)
    if corrected:
        xN, yN = xN + kx['cN'], yN + ky['cN']
    return x0, xN, y0, yN


def predictor_step(S, un, vn, un1, vn1, dt):
    """:232-337 (nu is not used by the reference here)."""
    Nx, Ny = S.nx, S.ny
    Dx, Dy = S.Dx[1:-1, 1:-1], S.Dy[1:-1, 1:-1]
    Dxx, Dyy = S.Dx_sqr[1:-1, 1:-1], S.Dy_sqr[1:-1, 1:-1]
    _un, _un1, _vn, _vn1 = un[1:-1, 1:-1], un1[1:-1, 1:-1], vn[1:-1, 1:-1], vn1[1:-1, 1:-1]

    def F(f, f1, name):
        if S.corrected:
            # every derivative sees the boundary values (folded operators + the g-dependent constants); the implicit side's
            # constants dt (cxx + cyy) move to the right-hand side.  Identical to the branch below for homogeneous Dirichlet data.
            o, h = S.folded[name], S.helm[name]
            dxf = lambda a: o['Dx'] @ a + o['cx'][:, None]
            dyf = lambda a: a @ o['Dy'].T + o['cy'][None, :]
            lap = lambda a: h['Mx'] @ a + a @ h['My'].T + o['cxx'][:, None] + o['cyy'][None, :]
            return (2 * f - 3 * dt * (_un * dxf(f) + _vn * dyf(f)) + dt * (_un1 * dxf(f1) + _vn1 * dyf(f1)) + dt * lap(f)
                    + dt * (o['cxx'][:, None] + o['cyy'][None, :]))
        return (2 * f - 3 * dt * (_un * (Dx @ f) + _vn * (f @ Dy.T)) +
                dt * (_un1 * (Dx @ f1) + _vn1 * (f1 @ Dy.T)) +
                dt * (Dxx @ f + f @ Dyy.T))

    out = []
    for name, f, f1 in (('u', _un, _un1), ('v', _vn, _vn1)):
        h = S.helm[name]
        Ht = h['P_inv'] @ F(f, f1, name)
        Hh = Ht @ h['Q_inv'].T
        hat = Hh / (2. - dt * h['lx'][:, None].repeat(Nx - 2, axis=1) -
                    dt * h['ly'][:, None].repeat(Ny - 2, axis=1).T)
        sol = h['P'] @ (hat @ h['Q'].T)
        x0, xN, y0, yN = _boundary_values(sol, S.bc[name], *S.k[name], corrected=S.corrected)
        full = np.zeros((Nx, Ny), dtype=sol.dtype)
        full[1:-1, 1:-1] = sol
        full[0, 1:-1], full[-1, 1:-1] = x0, xN
        full[1:-1, 0], full[1:-1, -1] = y0, yN
        out.append(full)
    return out[0], out[1]


def correction_step_corrected(S, ui, vi, p, dt, rho):
    """The projection the reference's :339-383 is after, for Setup(corrected=True) -- an option of the build:
      * the divergence of u* at the interior nodes uses the boundary values u* actually carries (full rows of D), not the
        Dirichlet data paired with the wrong ends (:353-361);
      * the pressure operator DxDPx Q + Q DyDPy^T has the constant pressure in its null space (lambda_x0 + lambda_y0 = 0, which is
        what sends the reference's Q to 1e17): that one mode is projected out;
      * the velocity update subtracts the pressure GRADIENT, (dt / rho) DPx Q, where the reference subtracts DxDPx Q (:378-381).
    Then the interior divergence of the corrected field vanishes (up to the incompatible constant mode of the data)."""
    Nx, Ny = S.nx, S.ny
    div = S.Dx[1:-1, :] @ ui[:, 1:-1] + vi[1:-1, :] @ S.Dy[1:-1, :].T
    Hh = S.PP_inv @ (rho / dt * div) @ S.PQ_inv.T
    lam = S.lpx[:, None].repeat(Nx - 2, axis=1) + S.lpy[:, None].repeat(Ny - 2, axis=1).T
    null = np.abs(lam) <= 1e-9 * np.abs(lam).max()
    Qh = np.where(null, 0.0, Hh / np.where(null, 1.0, lam))
    Q = S.PP @ (Qh @ S.PQ.T)
    u1, v1, p1 = ui.copy(), vi.copy(), p.copy()
    u1[1:-1, 1:-1] = u1[1:-1, 1:-1] - S.DPx @ Q * dt / rho
    v1[1:-1, 1:-1] = v1[1:-1, 1:-1] - Q @ S.DPy.T * dt / rho
    p1[1:-1, 1:-1] = Q
    return u1, v1, p1


def correction_step(S, ui, vi, p, dt, rho):
    """:339-383"""
    Nx, Ny = S.nx, S.ny
    gu, gv = S.bc['u'], S.bc['v']
    u_tau = np.stack([np.ones(Ny - 2) * gu['g_minus_x'], np.ones(Ny - 2) * gu['g_plus_x']])
    v_tau = np.stack([np.ones(Nx - 2) * gv['g_minus_y'], np.ones(Nx - 2) * gv['g_plus_y']]).T
    Dx_bar = np.stack([S.Dx[1:-1, 0], S.Dx[1:-1, -1]]).T
    Dy_bar = np.stack([S.Dy[1:-1, 0], S.Dy[1:-1, -1]]).T
    Sm = -(Dx_bar @ u_tau + v_tau @ Dy_bar.T)
    H = -rho / dt * (Sm - S.Dx[1:-1, 1:-1] @ ui[1:-1, 1:-1] - vi[1:-1, 1:-1] @ S.Dy[1:-1, 1:-1].T)
    Ht = S.PP_inv @ H
    Hh = Ht @ S.PQ_inv.T
    Qh = Hh / (S.lpx[:, None].repeat(Nx - 2, axis=1) + S.lpy[:, None].repeat(Ny - 2, axis=1).T)
    Q = S.PP @ (Qh @ S.PQ.T)
    u1, v1, p1 = ui.copy(), vi.copy(), p.copy()
    u1[1:-1, 1:-1] = u1[1:-1, 1:-1] - S.DxDPx @ Q * dt / rho
    v1[1:-1, 1:-1] = v1[1:-1, 1:-1] - Q @ S.DyDPy.T * dt / rho
    p1[1:-1, 1:-1] = Q
    return u1, v1, p1
