"""mspl_amd -- MI355X-native (gfx950) implementation of MSPL's segmentation + multi-source
pseudo-label hot path behind the reference's Python call surface.

Importing this package loads mspl_amd/lib/libmspl_hip.so (hand-written HIP kernels behind the C ABI of
include/mspl_hip.h); it raises ImportError if the library has not been built -- there is no fallback.
"""
from . import _native  # noqa: F401  (fails loudly when the HIP library is missing)
from . import autograd, dist, evaluation, layers, models, ops, training, uest  # noqa: F401
from .dropin import install_dropin  # noqa: F401

__version__ = '0.1'
