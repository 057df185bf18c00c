"""nns -- MI355X-native residual engine behind the call surface of mhw32/neural-navier-stokes.

Host side: Python mirrors of the reference classes (``nns.boundary``, ``nns.chorin_fd``,
``nns.direct_fd``, ``nns.periodic`` ...) that keep fields resident on the GPU and call the
hand-written HIP kernels of ``csrc/`` through the C ABI of ``include/nns.h`` (``nns._lib`` /
``nns.ops``).  ``src/`` next to this package re-exports them under the reference's import paths
(``from src.chorin_fd.simulate import NavierStokesSystem``).
"""
from . import _lib  # noqa: F401

__version__ = '0.1'
