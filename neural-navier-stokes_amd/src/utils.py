"""Reference path ``src/utils.py`` -> nns.utils."""
from nns.utils import numpy_to_torch, spatial_coarsen  # noqa: F401
