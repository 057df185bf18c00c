"""Reference path ``src/neural_spectral/spectral_ode2.py`` -> nns.neural_spectral.spectral_ode2."""
from nns.neural_spectral.spectral_ode2 import *  # noqa: F401,F403
from nns.neural_spectral.spectral_ode2 import ODEFunc, PDEFunc, AverageMeter  # noqa: F401
