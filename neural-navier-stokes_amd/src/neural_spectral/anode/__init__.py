"""Reference path ``src/neural_spectral/anode`` -> nns.neural_spectral.anode."""
from nns.neural_spectral.anode import odesolver, odesolver_adjoint  # noqa: F401
