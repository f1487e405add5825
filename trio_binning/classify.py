"""Drop-in alias of :mod:`trio_binning_amd.classify` (reference module: src/trio_binning/classify.py)."""
import sys as _sys

import trio_binning_amd.classify as _impl

_sys.modules[__name__] = _impl
