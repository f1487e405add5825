"""oracle — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's classify-by-kmers algorithm (``kmers_oracle.c``) plus,
when it has been built in the dev container, the real reference compiled from its own
source (``_ref/kmers_ref.so``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package.  Nothing under
``trio_binning_amd/`` imports it, and the product fails loudly without its HIP library
instead of falling back to anything here.
"""
from .binding import (  # noqa: F401
    Oracle,
    RefLib,
    build,
    have_ref,
    load,
    load_ref,
)
