"""Drop-in alias of :mod:`trio_binning_amd.seq` (reference module: src/trio_binning/seq.py)."""
import sys as _sys

import trio_binning_amd.seq as _impl

_sys.modules[__name__] = _impl
