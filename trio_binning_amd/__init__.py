"""trio_binning_amd — MI355X-native ``classify-by-kmers``.

Host-side mirror of the reference's hot-path modules (``kmers``, ``seq``,
``classify_by_kmers``) over the C-ABI of ``libtbk_hip.so`` (include/tbk.h): hand-written
HIP for gfx950.  Importing ``kmers`` needs the built library and raises ``ImportError``
without it; computing needs a visible MI355X.  There is no CPU fallback.
"""
__version__ = "0.1.0"
