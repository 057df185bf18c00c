"""Reference path ``src/neural_spectral/spectral_ode.py`` -> nns.neural_spectral.spectral_ode."""
from nns.neural_spectral.spectral_ode import *  # noqa: F401,F403
from nns.neural_spectral.spectral_ode import ODEFunc, PDEFunc, AverageMeter  # noqa: F401
