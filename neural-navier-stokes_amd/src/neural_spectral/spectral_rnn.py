"""Reference path ``src/neural_spectral/spectral_rnn.py`` -> nns.neural_spectral.spectral_rnn (GRU coefficient dynamics)."""
from nns.neural_spectral.spectral_rnn import *  # noqa: F401,F403
from nns.neural_spectral.spectral_rnn import PDEFunc, AverageMeter  # noqa: F401
