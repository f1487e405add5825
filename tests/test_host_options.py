"""tbk_options (include/tbk.h): how a classifier is built is an argument, not the environment.  The reference configures
itself through argparse alone (classify_by_kmers.py:14-54).  Host code only: defaults, the environment overlay the
command-line tools use as their fallback, and that nothing under tbk_classifier_create* / tbk_pipeline_create* reads the
environment.  (Two classifiers with different pinned layouts built concurrently: tests/test_gpu_sweep.py.)"""
import ctypes as C
import os
import re

from conftest import ROOT


def test_defaults_and_the_environment_overlay(built, monkeypatch):
    from trio_binning_amd import _lib, kmers

    o = kmers.Options()
    assert o.c.size == C.sizeof(_lib.tbk_options)
    assert (o.c.short_keys, o.c.entries, o.c.wide_entries, o.c.front, o.c.mod_sampling, o.c.span3, o.c.guests) == (-1,) * 7
    assert (o.c.minimizer_w, o.c.minimizer_m, o.c.two_read_kernel, o.c.packed_h2d, o.c.short_line_cap) == (-1, 0, 1, 1, 32)
    assert (o.c.table_load, o.c.entry_load, o.c.wentry_load, o.c.short_load) == (0.0, 0.5, 0.25, 2.3)
    assert (o.c.clustered, o.c.behind_front, o.c.plainly_clustered, o.c.entry_min_ratio) == (0.003, 0.05, 0.12, 1.5)
    assert o.c.memory_budget_bytes == 0 and o.c.slice_bases == 384 << 20
    assert repr(o) == "Options()"
    for var in list(os.environ):
        if var.startswith("TBK_"):
            monkeypatch.delenv(var)
    assert repr(kmers.Options.from_env()) == "Options()"
    monkeypatch.setenv("TBK_ENTRY", "1")
    monkeypatch.setenv("TBK_SHORT", "0")
    monkeypatch.setenv("TBK_ENTRY_LOAD", "0.64")
    monkeypatch.setenv("TBK_MINIMIZER_W", "5")
    monkeypatch.setenv("TBK_MEMORY_BUDGET", str(64 << 30))
    monkeypatch.setenv("TBK_SLICE_BASES", "2048")
    e = kmers.Options.from_env(front=0)
    assert (e.c.entries, e.c.short_keys, e.c.front, e.c.minimizer_w, e.c.entry_load, e.c.memory_budget_bytes, e.c.slice_bases) == (1, 0, 0, 5, 0.64, 64 << 30, 2048)
    assert e.c.wide_entries == -1 and e.c.short_load == 2.3
    assert repr(kmers.Options.layout("wide_entries", wentry_load=0.2)) == "Options(entries=1, wide_entries=1, wentry_load=0.2)"
    try:
        kmers.Options(no_such_field=1)
    except TypeError:
        pass
    else:
        raise AssertionError("an unknown option must be refused")


def test_the_constructors_do_not_read_the_environment():
    """Source-level guard: no getenv between tbk_classifier_create_opts and the end of the multi-device constructor, nor in
    the functions they call to build a table; tbk_pipeline_create* only forward."""
    src = open(os.path.join(ROOT, "trio_binning_amd", "csrc", "tbk_host.cpp")).read()
    for name in ("build_pair_table", "build_entry_table", "build_short_table", "classifier_streams", "tbk_classifier_create_opts", "tbk_classifier_replicate",
                 "tbk_classifier_create_multi_opts"):
        m = re.search(r"^(?:static |extern \"C\" )int " + name + r"\(.*?^}", src, re.S | re.M)
        assert m, name
        assert "getenv" not in m.group(0) and "env_double" not in m.group(0), name
    pipe = open(os.path.join(ROOT, "trio_binning_amd", "csrc", "tbk_pipeline.cpp")).read()
    m = re.search(r"^extern \"C\" int tbk_pipeline_create_opts\(.*?^}", pipe, re.S | re.M)
    assert m and "getenv" not in m.group(0)
    kernels = open(os.path.join(ROOT, "trio_binning_amd", "csrc", "tbk_kernels.hip")).read()
    assert "getenv" not in kernels
