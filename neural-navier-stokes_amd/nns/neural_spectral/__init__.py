"""neural_spectral field predictor: mirror of the reference's ``src/neural_spectral`` package
(``spectral_ode.py``, ``spectral_ode2.py``, ``anode/``) on fused HIP kernels (csrc/neural_kernels.hip)."""
from .anode import odesolver, odesolver_adjoint  # noqa: F401
