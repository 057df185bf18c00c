"""Reference path ``src/direct_fd/simulate.py`` -> nns.direct_fd."""
from nns.direct_fd import NavierStokesSystem  # noqa: F401
