"""Oracle (test infrastructure only): "direct" finite-difference Navier-Stokes (cavity).

NumPy restatement of ``src/direct_fd/simulate.py`` of the reference.  NOTE the axis convention
differs from chorin_fd: here **axis 1 is x** (``u[1:-1, 2:] - u[1:-1, 0:-2]`` is divided by
2*dx, src/direct_fd/simulate.py:60) and axis 0 is y.  dx = 2/(nx-1), dy = 2/(ny-1) (:53).
All functions accept leading batch axes.
"""
import numpy as np

from .boundary import apply_bc_list

_c = (Ellipsis, slice(1, -1), slice(1, -1))
_jp = (Ellipsis, slice(1, -1), slice(2, None))     # [i, j+1]
_jm = (Ellipsis, slice(1, -1), slice(0, -2))       # [i, j-1]
_ip = (Ellipsis, slice(2, None), slice(1, -1))     # [i+1, j]
_im = (Ellipsis, slice(0, -2), slice(1, -1))       # [i-1, j]


def build_up_b(u, v, dt, dx, dy, rho):
    """src/direct_fd/simulate.py:56-66."""
    b = np.zeros_like(u)
    b[_c] = (rho * (1 / dt *
                    ((u[_jp] - u[_jm]) / (2 * dx) +
                     (v[_ip] - v[_im]) / (2 * dy))) -
             ((u[_jp] - u[_jm]) / (2 * dx))**2 -
             2 * ((u[_ip] - u[_im]) / (2 * dy) *
                  (v[_jp] - v[_jm]) / (2 * dx)) -
             ((v[_ip] - v[_im]) / (2 * dy))**2)
    return b


def jacobi_sweep(p, b, dx, dy):
    """One Jacobi sweep, returns a NEW array (src/direct_fd/simulate.py:77-82)."""
    pn = p
    out = p.copy()
    out[_c] = (((pn[_jp] + pn[_jm]) * dy**2 +
                (pn[_ip] + pn[_im]) * dx**2) /
               (2 * (dx**2 + dy**2)) -
               dx**2 * dy**2 / (2 * (dx**2 + dy**2)) *
               b[_c])
    return out


def pressure_poisson(p, b, p_bc, dx, dy, nit):
    """src/direct_fd/simulate.py:68-88: exactly ``nit`` sweeps, p BCs after every sweep.
    Mutates ``p`` in place (as the reference) and returns it."""
    for _ in range(nit):
        p[...] = jacobi_sweep(p, b, dx, dy)
        apply_bc_list(p, p_bc)
    return p


def momentum_update(un, vn, p, dt, dx, dy, rho, nu):
    """The u, v update of src/direct_fd/simulate.py:98-118 (returns new arrays; edges copied)."""
    u, v = un.copy(), vn.copy()
    u[_c] = (un[_c] -
             un[_c] * dt / dx *
             (un[_c] - un[_jm]) -
             vn[_c] * dt / dy *
             (un[_c] - un[_im]) -
             dt / (2 * rho * dx) * (p[_jp] - p[_jm]) +
             nu * (dt / dx**2 *
                   (un[_jp] - 2 * un[_c] + un[_jm]) +
                   dt / dy**2 *
                   (un[_ip] - 2 * un[_c] + un[_im])))
    v[_c] = (vn[_c] -
             un[_c] * dt / dx *
             (vn[_c] - vn[_jm]) -
             vn[_c] * dt / dy *
             (vn[_c] - vn[_im]) -
             dt / (2 * rho * dy) * (p[_ip] - p[_im]) +
             nu * (dt / dx**2 *
                   (vn[_jp] - 2 * vn[_c] + vn[_jm]) +
                   dt / dy**2 *
                   (vn[_ip] - 2 * vn[_c] + vn[_im])))
    return u, v


def step(u, v, p, u_bc, v_bc, p_bc, dt, dx, dy, rho, nu, nit):
    """src/direct_fd/simulate.py:90-127.  Mutates u, v, p in place (as the reference)."""
    b = build_up_b(u, v, dt, dx, dy, rho)
    pressure_poisson(p, b, p_bc, dx, dy, nit)
    un, vn = momentum_update(u, v, p, dt, dx, dy, rho, nu)
    u[...] = un
    v[...] = vn
    apply_bc_list(u, u_bc)
    apply_bc_list(v, v_bc)
    return u, v, p


def simulate(u_ic, v_ic, p_ic, u_bc, v_bc, p_bc, nt, nit, dt, rho, nu):
    """src/direct_fd/simulate.py:129-144 (the ICs are mutated, as in the reference)."""
    nx, ny = u_ic.shape
    dx, dy = 2. / (nx - 1), 2. / (ny - 1)
    u, v, p = u_ic, v_ic, p_ic
    us, vs, ps = [], [], []
    for _ in range(nt):
        u, v, p = step(u, v, p, u_bc, v_bc, p_bc, dt, dx, dy, rho, nu, nit)
        us.append(u.copy()), vs.append(v.copy()), ps.append(p.copy())
    return np.stack(us), np.stack(vs), np.stack(ps)
