"""Reference path ``src/neural_spectral/rnn.py`` -> nns.neural_spectral.rnn (black-box GRU next-frame baseline)."""
from nns.neural_spectral.rnn import *  # noqa: F401,F403
from nns.neural_spectral.rnn import RNN  # noqa: F401
