"""Drop-in alias of :mod:`trio_binning_amd.find_unique_kmers` (reference module: src/trio_binning/find_unique_kmers.py)."""
import sys as _sys

import trio_binning_amd.find_unique_kmers as _impl

_sys.modules[__name__] = _impl
