"""Drop-in name: ``from trio_binning import kmers, seq`` and
``trio_binning.classify_by_kmers:main`` resolve to the MI355X implementation in
:mod:`trio_binning_amd`.

The reference ships ``trio_binning`` as the import name of its hot-path modules
(pyproject.toml:16-19); a user switching over keeps their imports.  Each submodule here
replaces itself with the ``trio_binning_amd`` module of the same name on import.
"""
