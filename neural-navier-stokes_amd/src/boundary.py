"""Reference path ``src/boundary.py`` -> nns.boundary."""
from nns.boundary import BaseBoundaryCondition, DirichletBoundaryCondition, NeumannBoundaryCondition  # noqa: F401
