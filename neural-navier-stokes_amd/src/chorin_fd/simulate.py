"""Reference path ``src/chorin_fd/simulate.py`` -> nns.chorin_fd."""
from nns.chorin_fd import NavierStokesSystem  # noqa: F401
