"""Reference path ``src/chorin_spectral/simulate.py`` -> nns.chorin_spectral."""
from nns.chorin_spectral import NavierStokesSystem, dup_vector_by_row, dup_vector_by_col  # noqa: F401
