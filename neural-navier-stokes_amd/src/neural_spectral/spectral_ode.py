"""Reference path ``src/neural_spectral/spectral_ode.py`` -> nns.neural_spectral.spectral_ode; as a script, the reference's
training driver (spectral_ode.py:140-224): same flags, same checkpoint dict keys, same extrapolation.npy."""
from nns.neural_spectral.spectral_ode import *  # noqa: F401,F403
from nns.neural_spectral.spectral_ode import ODEFunc, PDEFunc, AverageMeter  # noqa: F401

if __name__ == "__main__":
    from nns.neural_spectral.train import main
    main('spectral_ode')
