"""Reference path ``src/utils.py`` -> nns.utils."""
from nns.utils import numpy_to_torch, spatial_coarsen, AverageMeter, save_checkpoint, mean_squared_error  # noqa: F401
