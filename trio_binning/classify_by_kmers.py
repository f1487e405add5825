"""Drop-in alias of :mod:`trio_binning_amd.classify_by_kmers` (reference module: src/trio_binning/classify_by_kmers.py)."""
import sys as _sys

import trio_binning_amd.classify_by_kmers as _impl

_sys.modules[__name__] = _impl
