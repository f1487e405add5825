"""Drop-in alias of :mod:`trio_binning_amd.kmers` (reference module: src/trio_binning/kmers.py)."""
import sys as _sys

import trio_binning_amd.kmers as _impl

_sys.modules[__name__] = _impl
