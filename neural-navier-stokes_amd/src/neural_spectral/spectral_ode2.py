"""Reference path ``src/neural_spectral/spectral_ode2.py`` -> nns.neural_spectral.spectral_ode2; as a script, the reference's
training driver (spectral_ode.py:140-224): same flags, same checkpoint dict keys, same extrapolation.npy."""
from nns.neural_spectral.spectral_ode2 import *  # noqa: F401,F403
from nns.neural_spectral.spectral_ode2 import ODEFunc, PDEFunc, AverageMeter  # noqa: F401

if __name__ == "__main__":
    from nns.neural_spectral.train import main
    main('spectral_ode2')
