"""Reference path ``src/direct_fd/simulate.py`` -> nns.direct_fd; as a script, the reference's driver (:147-194):
writes ``./data.npz`` (u, v, p [nt, nx, ny] float64)."""
from nns.direct_fd import NavierStokesSystem  # noqa: F401

if __name__ == "__main__":
    import argparse
    import numpy as np
    from src.boundary import DirichletBoundaryCondition, NeumannBoundaryCondition

    ap = argparse.ArgumentParser()
    ap.add_argument('--nt', type=int, default=200)
    ap.add_argument('--nit', type=int, default=50)
    ap.add_argument('--nx', type=int, default=50)
    ap.add_argument('--ny', type=int, default=50)
    ap.add_argument('--dt', type=float, default=0.001)
    ap.add_argument('--rho', type=float, default=1)
    ap.add_argument('--nu', type=float, default=0.1)
    ap.add_argument('--out', default='./data.npz')
    a = ap.parse_args()
    nx, ny = a.nx, a.ny
    dx, dy = 2. / (nx - 1.), 2. / (ny - 1.)
    z = lambda: np.zeros((nx, ny))
    u_bc = [DirichletBoundaryCondition(0, 'left', dx, dy), DirichletBoundaryCondition(1, 'right', dx, dy),
            DirichletBoundaryCondition(0, 'top', dx, dy), DirichletBoundaryCondition(0, 'bottom', dx, dy)]
    v_bc = [DirichletBoundaryCondition(0, 'left', dx, dy), DirichletBoundaryCondition(0, 'right', dx, dy),
            DirichletBoundaryCondition(0, 'top', dx, dy), DirichletBoundaryCondition(0, 'bottom', dx, dy)]
    p_bc = [DirichletBoundaryCondition(0, 'top', dx, dy), NeumannBoundaryCondition(0, 'bottom', dx, dy),
            NeumannBoundaryCondition(0, 'left', dx, dy), NeumannBoundaryCondition(0, 'right', dx, dy)]
    system = NavierStokesSystem(z(), z(), z(), u_bc, v_bc, p_bc, nt=a.nt, nit=a.nit, nx=nx, ny=ny, dt=a.dt, rho=a.rho, nu=a.nu)
    u_data, v_data, p_data = system.simulate()
    np.savez(a.out, u=u_data, v=v_data, p=p_data)
