"""Reference path ``src/chorin_fd/simulate.py`` -> nns.chorin_fd; run as a script it is the reference's driver
(:274-324): lid-driven cavity, writes ``./data_{method}.npz`` with keys u, v, p ([nt, nx, ny] float64).  The
reference hard-codes its settings; they are the defaults of the flags below."""
from nns.chorin_fd import NavierStokesSystem  # noqa: F401

if __name__ == "__main__":
    import argparse
    import numpy as np
    from src.boundary import DirichletBoundaryCondition, NeumannBoundaryCondition

    ap = argparse.ArgumentParser()
    ap.add_argument('--nt', type=int, default=200)            # number of timesteps (:278)
    ap.add_argument('--nit', type=int, default=200)           # iterations for the elliptic pressure eqn (:279)
    ap.add_argument('--nx', type=int, default=51)
    ap.add_argument('--ny', type=int, default=51)
    ap.add_argument('--dt', type=float, default=0.001)
    ap.add_argument('--rho', type=float, default=1)
    ap.add_argument('--nu', type=float, default=0.1)
    ap.add_argument('--beta', type=float, default=1.25)
    ap.add_argument('--method', default='semi_implicit', choices=['semi_implicit', 'explicit'])   # (:287)
    ap.add_argument('--out', default=None, help='default ./data_{method}.npz (:324)')
    a = ap.parse_args()
    nx, ny = a.nx, a.ny
    dx, dy = 2. / (nx - 1.), 2. / (ny - 1.)
    u_ic, v_ic, p_ic = np.zeros((nx, ny)), np.zeros((nx, ny)), np.zeros((nx, ny))
    u_bc = [DirichletBoundaryCondition(0, 'left', dx, dy), DirichletBoundaryCondition(1, 'right', dx, dy),
            DirichletBoundaryCondition(0, 'top', dx, dy), DirichletBoundaryCondition(0, 'bottom', dx, dy)]
    v_bc = [DirichletBoundaryCondition(0, 'left', dx, dy), DirichletBoundaryCondition(0, 'right', dx, dy),
            DirichletBoundaryCondition(0, 'top', dx, dy), DirichletBoundaryCondition(0, 'bottom', dx, dy)]
    p_bc = [DirichletBoundaryCondition(0, 'top', dx, dy), NeumannBoundaryCondition(0, 'bottom', dx, dy),
            NeumannBoundaryCondition(0, 'left', dx, dy), NeumannBoundaryCondition(0, 'right', dx, dy)]
    system = NavierStokesSystem(u_ic, v_ic, p_ic, u_bc, v_bc, p_bc, nt=a.nt, nit=a.nit, nx=nx, ny=ny, dt=a.dt,
                                rho=a.rho, nu=a.nu, beta=a.beta, method=a.method)
    u_data, v_data, p_data = system.simulate()
    np.savez(a.out or './data_{}.npz'.format(a.method), u=u_data, v=v_data, p=p_data)
