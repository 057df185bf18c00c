"""Oracle (test infrastructure only): Chorin projection, finite differences.

NumPy restatement of ``src/chorin_fd/simulate.py`` of the reference, operator by operator,
keeping the reference's operation order so that float64 results agree to rounding (the
vectorised stencils are bitwise equal; the SOR sweep is evaluated in anti-diagonal wavefront
order, which is bitwise equal to the reference's lexicographic in-place double loop because
every point still sees exactly the same already-updated / not-yet-updated neighbours).

Conventions (reference): fields are C-contiguous [nx, ny]; axis 0 = x ('left'/'right'),
axis 1 = y ('bottom'/'top'); dx = 2/(nx-1), dy = 2/(ny-1) (src/chorin_fd/simulate.py:58).
All functions accept leading batch axes ([..., nx, ny]).

Reference quirks reproduced on purpose (SURVEY.md section 7 "Reference bugs are the spec"):
  * explicit predictor: the "y" advection term uses the axis-0 difference (:74,:76,:83,:85);
  * semi-implicit predictor: ``2 / nu * dx**2`` parses as (2/nu)*dx^2 (:108,:134) and the second
    ADI solve acts along axis 0 again (:159,:165), so nx == ny is required;
  * SOR: at most nit-1 sweeps because ``it`` starts at 1 (:183,:190);
  * correction: rho is not used (:207-208).
"""
import numpy as np

from .boundary import apply_bc_list

SOR_TOL = 5e-6          # src/chorin_fd/simulate.py:183


def grid_spacing(nx, ny):
    """src/chorin_fd/simulate.py:58"""
    return 2. / (nx - 1), 2. / (ny - 1)


def explicit_predictor(u, v, u1, v1, dt, dx, dy, nu):
    """AB2 advection + AB2 diffusion.  src/chorin_fd/simulate.py:63-91."""
    un, vn, un1, vn1 = u, v, u1, v1
    ui, vi = u.copy(), v.copy()
    c = (Ellipsis, slice(1, -1), slice(1, -1))
    xp = (Ellipsis, slice(2, None), slice(1, -1))
    xm = (Ellipsis, slice(None, -2), slice(1, -1))
    yp = (Ellipsis, slice(1, -1), slice(2, None))
    ym = (Ellipsis, slice(1, -1), slice(None, -2))

    def adv(a, b, f):        # a * d0x(f)/(2dx) + b * d0x(f)/(2dy)   (x-difference twice: quirk)
        return a[c] * (f[xp] - f[xm]) / (2 * dx) + b[c] * (f[xp] - f[xm]) / (2 * dy)

    def lap(f):
        return ((f[xp] - 2 * f[c] + f[xm]) / dx**2 + (f[yp] - 2 * f[c] + f[ym]) / dy**2)

    ui[c] = un[c] - dt * (3 / 2. * adv(un, vn, un) - 1 / 2. * adv(un1, vn1, un1)) \
        + dt * nu * (3 / 2. * lap(un) - 1 / 2. * lap(un1))
    vi[c] = vn[c] - dt * (3 / 2. * adv(un, vn, vn) - 1 / 2. * adv(un1, vn1, vn1)) \
        + dt * nu * (3 / 2. * lap(vn) - 1 / 2. * lap(vn1))
    return ui, vi


def explicit_predictor_corrected(u, v, u1, v1, dt, dx, dy, nu):
    """SURVEY.md section 8 (f) rank 3, "fixed y-advection": the explicit predictor with v d/dy taken along y
    (the reference differences along x twice, src/chorin_fd/simulate.py:73-76,:82-85).  An OPTION of the build,
    not reference behaviour: pinned analytically (tests/test_oracle_golden.py: a field that varies only in y is
    advected by v, which the reference's form cannot do)."""
    ui, vi = u.copy(), v.copy()
    c = (Ellipsis, slice(1, -1), slice(1, -1))
    xp = (Ellipsis, slice(2, None), slice(1, -1))
    xm = (Ellipsis, slice(None, -2), slice(1, -1))
    yp = (Ellipsis, slice(1, -1), slice(2, None))
    ym = (Ellipsis, slice(1, -1), slice(None, -2))

    def adv(a, b, f):
        return a[c] * (f[xp] - f[xm]) / (2 * dx) + b[c] * (f[yp] - f[ym]) / (2 * dy)

    def lap(f):
        return ((f[xp] - 2 * f[c] + f[xm]) / dx**2 + (f[yp] - 2 * f[c] + f[ym]) / dy**2)

    ui[c] = u[c] - dt * (3 / 2. * adv(u, v, u) - 1 / 2. * adv(u1, v1, u1)) + dt * nu * (3 / 2. * lap(u) - 1 / 2. * lap(u1))
    vi[c] = v[c] - dt * (3 / 2. * adv(u, v, v) - 1 / 2. * adv(u1, v1, v1)) + dt * nu * (3 / 2. * lap(v) - 1 / 2. * lap(v1))
    return ui, vi


def sor_sweep_redblack(p, C, dx, dy, beta):
    """One red-black SOR sweep (same update formula as :193-196): first the points with (i + j) even, then the odd
    ones.  Within a colour every update reads only the other colour, so the half-sweep is order-independent --
    fully parallel, and shardable across GPUs with one halo exchange per half-sweep."""
    nx, ny = p.shape[-2], p.shape[-1]
    dx2, dy2 = dx**2, dy**2
    den = (2 * dx**2 + 2 * dy**2)
    I, J = np.meshgrid(np.arange(1, nx - 1), np.arange(1, ny - 1), indexing='ij')
    for colour in (0, 1):
        m = ((I + J) % 2) == colour
        i, j = I[m], J[m]
        p[..., i, j] = (beta * (dy2 * p[..., i + 1, j] + dy2 * p[..., i - 1, j] +
                                dx2 * p[..., i, j + 1] + dx2 * p[..., i, j - 1] -
                                C[..., i, j]) / den + (1 - beta) * p[..., i, j])
    return p


def get_pressure_redblack(ui, vi, p, dt, dx, dy, rho, beta, nit, tol=SOR_TOL, return_info=False):
    """get_pressure with red-black sweeps: same right-hand side, relaxation factor, stopping rule and sweep cap
    (:184-202); the iterates differ from the lexicographic order (an option, not reference behaviour)."""
    assert p.ndim == 2
    err, it = 1, 1
    pPrev = p.copy()
    C = pressure_rhs(ui, vi, dt, dx, dy, rho)
    sweeps = 0
    while (err > tol) and (it < nit):
        sor_sweep_redblack(p, C, dx, dy, beta)
        err = np.max(np.abs(p - pPrev))
        pPrev = p.copy()
        it += 1
        sweeps += 1
    if return_info:
        return p, (sweeps, float(err))
    return p


def thomas_const(lo, di, up, rhs):
    """Solve tridiag(lo, di, up) X = rhs along axis -2 of rhs ([..., n, m]) for a constant-
    coefficient tridiagonal matrix.  No pivoting: identical arithmetic to an LU of the dense
    matrix the reference builds (src/chorin_fd/simulate.py:105-121,:137) because the matrix
    is strictly diagonally dominant (|di| > |lo| + |up| is not needed for the pivot choice,
    only |di'| >= |lo|, which holds for the reference's positive dt, nu)."""
    n = rhs.shape[-2]
    cp = np.empty(n, dtype=rhs.dtype)         # modified diagonal
    y = np.empty_like(rhs)
    cp[0] = di
    y[..., 0, :] = rhs[..., 0, :]
    for i in range(1, n):
        l = lo / cp[i - 1]
        cp[i] = di - l * up
        y[..., i, :] = rhs[..., i, :] - l * y[..., i - 1, :]
    x = np.empty_like(rhs)
    x[..., n - 1, :] = y[..., n - 1, :] / cp[n - 1]
    for i in range(n - 2, -1, -1):
        x[..., i, :] = (y[..., i, :] - up * x[..., i + 1, :]) / cp[i]
    return x


def semi_implicit_predictor(u, v, u1, v1, dt, dx, dy, nu, column_slab=False):
    """AB2 advection (correct axes) + Crank-Nicolson diffusion by ADI.
    src/chorin_fd/simulate.py:93-167.  Both solves act along axis 0 (quirk).
    column_slab=True: the arrays are a column slab [nx, nyl] of a square grid (tests of the sharded step): the same
    arithmetic, the systems along axis 0 have the full size nx - 2; only the squareness assert is waived."""
    assert column_slab or u.shape[-1] == u.shape[-2], "reference semi_implicit needs nx == ny (:159,:165)"
    un, vn, un1, vn1 = u, v, u1, v1
    ut, vt = u.copy(), v.copy()
    ui, vi = u.copy(), v.copy()
    c = (Ellipsis, slice(1, -1), slice(1, -1))
    xp = (Ellipsis, slice(2, None), slice(1, -1))
    xm = (Ellipsis, slice(None, -2), slice(1, -1))
    yp = (Ellipsis, slice(1, -1), slice(2, None))
    ym = (Ellipsis, slice(1, -1), slice(None, -2))

    a_diag = 2 / nu * dx**2 + 2 * dt          # :108
    b_diag = 2 / nu * dy**2 + 2 * dt          # :117

    def H(a, b, f):                           # :126-129
        return a[c] * (f[xp] - f[xm]) / (2 * dx) + b[c] * (f[yp] - f[ym]) / (2 * dy)

    def first(f, f1):                         # :131-137
        C1 = dt / 2. * (3 * H(un, vn, f) - H(un1, vn1, f1))
        C2 = dt * nu * ((f[xp] - 2 * f[c] + f[xm]) / dx**2 + (f[yp] - 2 * f[c] + f[ym]) / dy**2)
        C = 2 / nu * dx**2 * (C1 + C2)
        return thomas_const(-dt, a_diag, -dt, C)

    ut[c] = first(un, un1)
    vt[c] = first(vn, vn1)

    def second(ft, f):                        # :157-159
        S = (2 / nu * dy**2 * (ft[c] + f[c]) - dt * (f[yp] - 2 * f[c] + f[ym]))
        return thomas_const(-dt, b_diag, -dt, S)

    ui[c] = second(ut, un)
    vi[c] = second(vt, vn)
    return ui, vi


def semi_implicit_predictor_corrected(u, v, u1, v1, dt, dx, dy, nu):
    """SURVEY.md section 8 (f) rank 3, "true y-direction ADI": semi_implicit_predictor with its second solve along
    axis 1 (rows), as an alternating-direction scheme intends (:157-165 solve along axis 0 again).  nx != ny allowed.
    An option of the build, not reference behaviour; pinned by the transposition identity in
    tests/test_oracle_golden.py (for fields that depend on y only, the y-solve of the corrected scheme equals the
    reference's axis-0 solve applied to the transposed problem)."""
    un, vn, un1, vn1 = u, v, u1, v1
    ut, vt = u.copy(), v.copy()
    ui, vi = u.copy(), v.copy()
    c = (Ellipsis, slice(1, -1), slice(1, -1))
    xp = (Ellipsis, slice(2, None), slice(1, -1))
    xm = (Ellipsis, slice(None, -2), slice(1, -1))
    yp = (Ellipsis, slice(1, -1), slice(2, None))
    ym = (Ellipsis, slice(1, -1), slice(None, -2))
    a_diag = 2 / nu * dx**2 + 2 * dt
    b_diag = 2 / nu * dy**2 + 2 * dt

    def H(a, b, f):
        return a[c] * (f[xp] - f[xm]) / (2 * dx) + b[c] * (f[yp] - f[ym]) / (2 * dy)

    def first(f, f1):
        C1 = dt / 2. * (3 * H(un, vn, f) - H(un1, vn1, f1))
        C2 = dt * nu * ((f[xp] - 2 * f[c] + f[xm]) / dx**2 + (f[yp] - 2 * f[c] + f[ym]) / dy**2)
        return thomas_const(-dt, a_diag, -dt, 2 / nu * dx**2 * (C1 + C2))

    ut[c] = first(un, un1)
    vt[c] = first(vn, vn1)

    def second(ft, f):
        S = (2 / nu * dy**2 * (ft[c] + f[c]) - dt * (f[yp] - 2 * f[c] + f[ym]))
        return np.swapaxes(thomas_const(-dt, b_diag, -dt, np.ascontiguousarray(np.swapaxes(S, -1, -2))), -1, -2)

    ui[c] = second(ut, un)
    vi[c] = second(vt, vn)
    return ui, vi


def pressure_rhs(ui, vi, dt, dx, dy, rho):
    """dx2dy2C of src/chorin_fd/simulate.py:186-188 (backward differences, zero on the edge)."""
    C = np.zeros_like(ui)
    c = (Ellipsis, slice(1, -1), slice(1, -1))
    C[c] = (dx * rho * dy**2 / dt * (ui[c] - ui[..., :-2, 1:-1]) +
            dy * rho * dx**2 / dt * (vi[c] - vi[..., 1:-1, :-2]))
    return C


def sor_sweep_wavefront(p, C, dx, dy, beta):
    """One in-place lexicographic SOR sweep (src/chorin_fd/simulate.py:191-196) evaluated by
    anti-diagonals d = i + j: every point of a diagonal depends only on diagonal d-1 (already
    updated this sweep) and d+1 (not yet updated) -- exactly what the i-outer/j-inner loop sees."""
    nx, ny = p.shape[-2], p.shape[-1]
    dx2, dy2 = dx**2, dy**2
    den = (2 * dx**2 + 2 * dy**2)
    for d in range(2, nx + ny - 3):
        i = np.arange(max(1, d - (ny - 2)), min(nx - 2, d - 1) + 1)
        j = d - i
        p[..., i, j] = (beta * (dy2 * p[..., i + 1, j] + dy2 * p[..., i - 1, j] +
                                dx2 * p[..., i, j + 1] + dx2 * p[..., i, j - 1] -
                                C[..., i, j]) / den + (1 - beta) * p[..., i, j])
    return p


def sor_sweep_lexicographic(p, C, dx, dy, beta):
    """The reference's literal double loop (:191-196); pure Python -- small grids only."""
    nx, ny = p.shape
    for i in range(1, nx - 1):
        for j in range(1, ny - 1):
            p[i, j] = (beta * (dy**2 * p[i + 1, j] + dy**2 * p[i - 1, j] +
                               dx**2 * p[i, j + 1] + dx**2 * p[i, j - 1] -
                               C[i, j]) / (2 * dx**2 + 2 * dy**2) +
                       (1 - beta) * p[i, j])
    return p


def get_pressure(ui, vi, p, dt, dx, dy, rho, beta, nit, tol=SOR_TOL, return_info=False):
    """src/chorin_fd/simulate.py:169-202.  Mutates and returns ``p`` (2-D only: the stopping
    test is per grid).  info = (sweeps done, last err)."""
    assert p.ndim == 2
    err, it = 1, 1
    pPrev = p.copy()
    C = pressure_rhs(ui, vi, dt, dx, dy, rho)
    sweeps = 0
    while (err > tol) and (it < nit):
        sor_sweep_wavefront(p, C, dx, dy, beta)
        err = np.max(np.abs(p - pPrev))
        pPrev = p.copy()
        it += 1
        sweeps += 1
    if return_info:
        return p, (sweeps, float(err))
    return p


def correction(ui, vi, p, dt, dx, dy):
    """src/chorin_fd/simulate.py:204-210 (rho unused, as in the reference)."""
    un1, vn1 = ui.copy(), vi.copy()
    c = (Ellipsis, slice(1, -1), slice(1, -1))
    un1[c] = ui[c] - dt / (2 * dx) * (p[..., 2:, 1:-1] - p[..., :-2, 1:-1])
    vn1[c] = vi[c] - dt / (2 * dy) * (p[..., 1:-1, 2:] - p[..., 1:-1, :-2])
    return un1, vn1


def step(un, vn, un1, vn1, p, u_bc, v_bc, p_bc, dt, dx, dy, rho, nu, beta, nit,
         method='semi_implicit', return_info=False, advection='reference', pressure_solver='sor'):
    """src/chorin_fd/simulate.py:212-234.  ``p`` is mutated (as in the reference).  advection / pressure_solver select
    the build's corrected options (section 8 (f) rank 3); the defaults are the reference."""
    if method == 'explicit':
        pred = explicit_predictor_corrected if advection == 'corrected' else explicit_predictor
        ui, vi = pred(un, vn, un1, vn1, dt, dx, dy, nu)
    elif method == 'semi_implicit':
        pred = semi_implicit_predictor_corrected if advection == 'corrected' else semi_implicit_predictor
        ui, vi = pred(un, vn, un1, vn1, dt, dx, dy, nu)
    else:
        raise Exception('method not recognized: {}'.format(method))
    apply_bc_list(ui, u_bc)
    apply_bc_list(vi, v_bc)
    solve = get_pressure_redblack if pressure_solver == 'redblack' else get_pressure
    p, info = solve(ui, vi, p, dt, dx, dy, rho, beta, nit, return_info=True)
    apply_bc_list(p, p_bc)
    u_new, v_new = correction(ui, vi, p, dt, dx, dy)
    if return_info:
        return u_new, v_new, p, info
    return u_new, v_new, p


def simulate(u_ic, v_ic, p_ic, u_bc, v_bc, p_bc, nt, nit, dt, rho, nu, beta, method, advection='reference', pressure_solver='sor'):
    """src/chorin_fd/simulate.py:236-271.  Returns stacked [nt, nx, ny] u, v, p."""
    nx, ny = u_ic.shape
    dx, dy = grid_spacing(nx, ny)
    u, v, p = u_ic.copy(), v_ic.copy(), p_ic.copy()
    apply_bc_list(u, u_bc)
    apply_bc_list(v, v_bc)
    apply_bc_list(p, p_bc)
    u1, v1 = u.copy(), v.copy()
    us, vs, ps = [], [], []
    for _ in range(nt):
        _u, _v, p = step(u, v, u1, v1, p, u_bc, v_bc, p_bc, dt, dx, dy, rho, nu, beta, nit, method,
                         advection=advection, pressure_solver=pressure_solver)
        u1, v1 = u, v
        u, v = _u, _v
        us.append(u.copy()), vs.append(v.copy()), ps.append(p.copy())
    return np.stack(us), np.stack(vs), np.stack(ps)
