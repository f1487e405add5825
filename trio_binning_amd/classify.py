"""Alias of :mod:`trio_binning_amd.classify_by_kmers`.

BASELINE.json's north star names a ``trio_binning.classify`` entry point; in the
reference the module is ``trio_binning.classify_by_kmers`` (pyproject.toml:18).  Both
names resolve to the same driver here.
"""
from .classify_by_kmers import (  # noqa: F401
    calculate_scaling_factors,
    main,
    output_extension,
    parse_args,
)

if __name__ == "__main__":
    main()
