"""The multi-device product path on what a one-GPU lease offers: two classifiers on device 0
(`TBK_DEVICES=0,0`: the table replicated device-to-device, two stream rings), batches dealt between
them, results in input order.  Counts, TSV and bins must equal the single-classifier run's and the
reference's recorded output."""
import ctypes as C
import gzip
import hashlib
import os
from unittest.mock import patch

import numpy as np
import pytest

from conftest import DATA, load_golden

pytestmark = pytest.mark.gpu


def _lists(tmp_path, k=21):
    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == k)
    fa, fb = tmp_path / "la.txt", tmp_path / "lb.txt"
    fa.write_text("".join(x + "\n" for x in v["list_a"]))
    fb.write_text("".join(x + "\n" for x in v["list_b"]))
    return v, str(fa), str(fb)


def test_replicated_tables_answer_alike(gpu, orc, tmp_path):
    from trio_binning_amd import _lib, kmers
    from trio_binning_amd._lib import check, lib

    v, fa, fb = _lists(tmp_path)
    a, b = kmers.HashSet.from_file(fa, 0), kmers.HashSet.from_file(fb, 0)
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    bases, offs = kmers.pack_reads(v["reads"])
    want = orc.count_batch(bases, offs, oa, ob)
    with kmers.MultiClassifier(a, b, [0, 0, 0]) as multi:
        assert multi.devices == [0, 0, 0] and multi.depth == 9
        st = multi.stats()
        assert st["devices"] == [0, 0, 0] and st["table_bytes_total"] == st["table_bytes"]   # three rings on one device share its table
        for i in range(3):              # every replica on its own (the pipeline is idle)
            part = multi._part(i)
            assert part.stats() == multi._part(0).stats()   # (one shared table)
            assert np.array_equal(part.classify_batch(bases, offs), want)
        # queued: twelve batches submitted before the first wait (depth 9 + one waiting per ring), ASCII and
        # packed ahead alike, waited for out of order
        cuts = [0, 10, 11, 40, 41, 41, 60, 75, 90, 100, 120, 149, len(v["reads"])]
        tickets = []
        for i, (lo, hi) in enumerate(zip(cuts, cuts[1:])):
            bb, oo = kmers.pack_reads(v["reads"][lo:hi])
            tickets.append((lo, hi, multi.submit(bb, oo) if i % 2 else multi.submit_packed(kmers.pack_bases(bb, oo))))
        with pytest.raises(_lib.TbkError):
            multi.submit(bases, offs)   # further ahead than depth + rings
        for lo, hi, t in reversed(tickets):
            assert np.array_equal(multi.wait(t), want[lo:hi]), (lo, hi)
        assert sum(multi.dealt) == 12, multi.dealt   # (how the feeders share them is a race; each ring has answered by itself above)
    # a replica of a replica, and the error paths of the C entry points
    with kmers.Classifier(a, b) as one:
        h = C.c_void_p()
        check(lib.tbk_classifier_replicate(one._h, 0, C.byref(h)))
        with kmers.Classifier(a, b, _handle=h.value) as two:
            assert two.stats() == one.stats() and np.array_equal(two.classify_batch(bases, offs), want)
        assert lib.tbk_classifier_replicate(one._h, 99, C.byref(h)) == _lib.TBK_ERR_INVALID
        out = (C.c_void_p * 2)()
        bad = (C.c_int * 2)(0, 99)
        assert lib.tbk_classifier_create_multi(a._h, b._h, bad, 2, out) == _lib.TBK_ERR_INVALID
        assert not out[0] and not out[1]
        assert lib.tbk_classifier_create_multi(a._h, b._h, bad, 0, out) == _lib.TBK_ERR_INVALID


# what two tables of the same lists must share whichever way the second was made; which key of a crowded bucket lies behind a
# front, and whether two neighbouring runs meet in one entry, depends on the order the racing inserts arrive in
GEOMETRY = ("entry_layout", "wide_entries", "short_keys", "full_keys", "shared_keys", "front_layout", "distinct_a", "distinct_b", "n_buckets", "table_bytes",
            "minimizer_w", "minimizer_m", "span_offset", "sampling_t")


def geometry(stats):
    return {name: stats[name] for name in GEOMETRY}


@pytest.mark.parametrize("how", ["built", "copied"])
@pytest.mark.parametrize("layout", ["key", "entry", "wide", "short", "full"])
def test_forced_replicas_are_real_tables_and_answer_alike(gpu, orc, tmp_path, monkeypatch, layout, how):
    """TBK_FORCE_REPLICA=1: every further ring on device 0 gets a table of its own, made by the code that makes a second GPU's -
    the fan-out of an 8-GPU node, executed on the one GPU there is.  `built` (the default): every replica gets the lists' keys and
    builds the table again, all replicas at once on their own host threads, in the layout and geometry the first build decided
    on (tbk_classifier_create_multi_opts; c/kmers.c:185-229 builds once and :245-268 only reads, so tables built independently
    from the same lines answer alike); `copied` (TBK_REPLICA_COPY=1): the finished table is copied, asynchronously.  In each
    layout (a replica of short keys carries the overflow table behind its lines): three tables in three places with one
    geometry, every one of them classifies like the oracle, alone and behind the pipeline's queue, every one was asked for
    every list line when the layout merges keys (or TBK_VERIFY_BUILD=1 says so); without the switch the rings share one table."""
    from trio_binning_amd import kmers

    v, fa, fb = _lists(tmp_path, 31 if layout in ("wide", "full") else 21)
    a, b = kmers.HashSet.from_file(fa, 0), kmers.HashSet.from_file(fb, 0)
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    bases, offs = kmers.pack_reads(v["reads"])
    want = orc.count_batch(bases, offs, oa, ob)
    n_lines = len(v["list_a"]) + len(v["list_b"])

    for var in ("TBK_FORCE_REPLICA", "TBK_REPLICA_COPY", "TBK_VERIFY_BUILD"):
        monkeypatch.delenv(var, raising=False)
    env = {"key": {"TBK_ENTRY": "0", "TBK_SHORT": "0", "TBK_FULL": "0"}, "entry": {"TBK_ENTRY": "1"}, "wide": {"TBK_ENTRY": "1", "TBK_ENTRY_WIDE": "1"},
           "short": {"TBK_SHORT": "1"}, "full": {"TBK_FULL": "1"}}[layout]
    for var, val in env.items():
        monkeypatch.setenv(var, val)
    merging = layout in ("entry", "wide")
    with kmers.MultiClassifier(a, b, [0, 0, 0]) as multi:
        ids = [multi._part(i).table_id() for i in range(3)]
        assert len({t for t, _ in ids}) == 1 and all(r == 0 for _, r in ids)
        st = multi._part(0).stats()
        assert st["entry_layout"] == merging and st["wide_entries"] == (layout == "wide") and st["short_keys"] == (layout == "short") and st["full_keys"] == (layout == "full"), st
        # the shared table was asked for every list line once; the rings that share it say so too
        assert [multi._part(i).verified()["lines"] for i in range(3)] == [n_lines if merging else 0] * 3
    monkeypatch.setenv("TBK_FORCE_REPLICA", "1")
    if how == "copied":
        monkeypatch.setenv("TBK_REPLICA_COPY", "1")
    if layout == "short":
        monkeypatch.setenv("TBK_VERIFY_BUILD", "1")   # (every layout on request)
    with kmers.MultiClassifier(a, b, [0, 0, 0]) as multi:
        ids = [multi._part(i).table_id() for i in range(3)]
        assert len({t for t, _ in ids}) == 3 and sorted(r for _, r in ids) == ([0, 2, 2] if how == "built" else [0, 1, 1]), ids
        for i in range(3):
            part = multi._part(i)
            assert geometry(part.stats()) == geometry(multi._part(0).stats())
            assert part.verified()["lines"] == (n_lines if merging or layout == "short" else 0), (i, part.verified())
            assert np.array_equal(part.classify_batch(bases, offs), want), i
            assert part.verify(a, b)["bad_lines"] == 0
        cuts = [0, 7, 30, 31, 64, 100, 149, len(v["reads"])]
        tickets = []
        for i, (lo, hi) in enumerate(zip(cuts, cuts[1:])):
            bb, oo = kmers.pack_reads(v["reads"][lo:hi])
            tickets.append((lo, hi, multi.submit(bb, oo) if i % 2 else multi.submit_packed(kmers.pack_bases(bb, oo))))
        for lo, hi, t in tickets:
            assert np.array_equal(multi.wait(t), want[lo:hi]), (lo, hi)
        assert sum(multi.dealt) == len(tickets), multi.dealt   # (which ring takes a batch is a race between the feeders; every replica has answered by itself above)
    monkeypatch.setenv("TBK_VERIFY_BUILD", "0")
    with kmers.MultiClassifier(a, b, [0, 0]) as multi:
        assert [multi._part(i).verified()["lines"] for i in range(2)] == [0, 0]
        assert np.array_equal(multi._part(1).classify_batch(bases, offs), want)


def test_device_numa_node_and_feeder_binding(gpu, tmp_path):
    """The device's NUMA node as the kernel reports it, and the pipeline's feeder bound to that node's CPUs (or left
    alone where the node is unknown: both are legal on a box, the record must be consistent)."""
    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    node = C.c_int(-2)
    check(lib.tbk_device_numa_node(0, C.byref(node)))
    assert node.value >= -1
    ncpu = lib.tbk_numa_node_cpus_(node.value, None, 0)
    v, fa, fb = _lists(tmp_path)
    a, b = kmers.HashSet.from_file(fa, 0), kmers.HashSet.from_file(fb, 0)
    bases, offs = kmers.pack_reads(v["reads"][:20])
    with kmers.MultiClassifier(a, b, [0, 0]) as multi:
        multi.wait(multi.submit(bases, offs))
        for slot in range(2):
            n, c = C.c_int(-2), C.c_int(-2)
            check(lib.tbk_pipeline_numa(multi._h, slot, C.byref(n), C.byref(c)))
            assert n.value == node.value
            assert 0 <= c.value <= max(ncpu, 0) if node.value >= 0 else c.value == 0
    assert lib.tbk_device_numa_node(99, C.byref(node)) != 0


@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
def test_cli_on_device_list_writes_the_same_bytes(gpu, capsys, tmp_path, monkeypatch, devices):
    import trio_binning_amd.classify_by_kmers as cbk

    v, fa, fb = _lists(tmp_path)
    fq = tmp_path / "reads21.fa"
    with open(fq, "w") as fh:
        for i, s in enumerate(v["reads"]):
            fh.write(f">r{i} some comment\n{s}\n")
    monkeypatch.setenv("TBK_DEVICES", devices)
    monkeypatch.setenv("TBK_STATS", "1")
    monkeypatch.setattr(cbk, "_BATCH_BASES", 300)
    monkeypatch.setattr(cbk, "_BATCH_READS", 5)
    od = tmp_path / "out"
    od.mkdir()
    with patch("sys.argv", ["classify-by-kmers", str(fq), fa, fb, "--haplotype-a-out-prefix", str(od / "hapA"),
                            "--haplotype-b-out-prefix", str(od / "hapB"), "--unclassified-out-prefix", str(od / "unclassified")]):
        cbk.main()
    out, err = capsys.readouterr()
    assert out == v["cli_stdout"]
    for fn, digest in v["cli_bins"].items():
        assert hashlib.sha256(gzip.open(od / fn, "rb").read()).hexdigest() == digest, fn
    assert '"devices": [%s]' % devices.replace(",", ", ") in err
