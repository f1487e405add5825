"""The multi-device product path without a GPU: the library's ``tbk_pipeline`` (``kmers.MultiClassifier``)
queues batches, one feeder thread per ring takes them and hands results back by ticket, and the native
classify loop (``tbk_classify_file``) on top of it writes TSV and bins in input order.  The rings are
stubs here, through the library's testing hook (they count with the oracle on a thread of their own and
finish out of step with one another); the real ones are exercised by tests/test_gpu_multi.py."""
import gzip
import hashlib
import os
import random
import threading
import time
from unittest.mock import patch

import numpy as np
import pytest

from conftest import DATA, load_golden


class StubClassifier:
    """Classifier-shaped: submit/submit_batch/wait with a ticket ring of `depth`, tickets finishing
    in submission order after a random delay, counts from `count_fn(bases, offsets)`."""

    depth = 3

    def __init__(self, device, count_fn, seed):
        self.device = device
        self.count_fn = count_fn
        self.rng = random.Random(seed)
        self.jobs = {}
        self.next_ticket = 1
        self.submitted = []
        self.closed = False

    def stats(self):
        return {"table_bytes": 1024, "n_buckets": 8}

    def submit(self, bases, offsets):
        assert len(self.jobs) < self.depth, "more batches in flight than the ring holds"
        bases, offsets = np.array(bases, copy=True), np.array(offsets, copy=True)
        t = self.next_ticket
        self.next_ticket += 1
        box = {}
        delay = self.rng.random() * 0.02

        def work():
            time.sleep(delay)
            box["counts"] = self.count_fn(bases, offsets)

        th = threading.Thread(target=work)
        th.start()
        self.jobs[t] = (th, box)
        self.submitted.append(int(offsets[-1]))
        return t

    def submit_batch(self, batch):
        arrays = batch.arrays()
        return self.submit(arrays[0], arrays[1])

    def wait(self, ticket):
        assert ticket == min(self.jobs), "a classifier's tickets are waited for in submission order here"
        th, box = self.jobs.pop(ticket)
        th.join()
        return box["counts"]

    def sync(self):
        pass

    def close(self):
        self.closed = True


def _lens(bases, offsets):
    return np.stack([np.diff(offsets).astype(np.int32), np.full(offsets.size - 1, bases.size, dtype=np.int32)], axis=1)


def test_pipeline_deals_and_maps_tickets(built):
    """The library's queue and feeder threads (tbk_pipeline, here over stub rings): every batch comes back
    under its own ticket whichever ring computed it, every ring takes batches, a ring never holds more than
    its depth, and running further ahead than depth + rings is refused."""
    from trio_binning_amd import _lib, kmers

    parts = [StubClassifier(d, _lens, d) for d in (0, 1, 2)]
    multi = kmers.MultiClassifier.from_classifiers(parts)
    assert multi.depth == 9 and multi.devices == [0, 1, 2]
    rng = np.random.default_rng(0)
    batches = []
    for i in range(60):
        lens = rng.integers(0, 50, int(rng.integers(1, 6)))
        offs = np.zeros(lens.size + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        batches.append((np.full(int(offs[-1]), 65, dtype=np.uint8), offs))
    pending, got = [], {}
    limit = multi.depth + len(parts)
    for i, (b, o) in enumerate(batches):
        if len(pending) == limit:
            with pytest.raises(_lib.TbkError):
                multi.submit(b, o)         # too far ahead of the waits
            j, t = pending.pop(0)
            got[j] = multi.wait(t)
        pending.append((i, multi.submit(b, o)))
    for j, t in reversed(pending):          # tickets may be waited for in any order
        got[j] = multi.wait(t)
    for i, (b, o) in enumerate(batches):
        assert np.array_equal(got[i], _lens(b, o)), i
    assert sum(multi.dealt) == 60 and min(multi.dealt) >= 5, multi.dealt
    with pytest.raises(_lib.TbkError):
        multi.wait(12345)
    multi.close()
    assert all(p.closed for p in parts)


def test_pipeline_reports_a_ring_failure_on_that_ticket(built):
    from trio_binning_amd import kmers

    class Broken(StubClassifier):
        def submit(self, bases, offsets):
            if int(offsets[-1]) == 13:
                raise RuntimeError("ring refused the batch")
            return super().submit(bases, offsets)

    multi = kmers.MultiClassifier.from_classifiers([Broken(0, _lens, 1), Broken(1, _lens, 2)])
    ok = (np.zeros(4, dtype=np.uint8), np.array([0, 4], dtype=np.uint64))
    bad = (np.zeros(13, dtype=np.uint8), np.array([0, 13], dtype=np.uint64))
    t1, t2, t3 = multi.submit(*ok), multi.submit(*bad), multi.submit(*ok)
    assert np.array_equal(multi.wait(t1), _lens(*ok))
    with pytest.raises(RuntimeError, match="refused"):
        multi.wait(t2)
    assert np.array_equal(multi.wait(t3), _lens(*ok))
    multi.close()


@pytest.mark.parametrize("n_devices", [2, 3])
def test_cli_on_several_devices_writes_input_order(built, orc, capsys, tmp_path, monkeypatch, n_devices):
    """classify-by-kmers over N per-device classifiers (stubs counting with the oracle, finishing out
    of step): stdout TSV and bins equal the reference's recorded output for the same inputs, i.e.
    what one device writes."""
    import trio_binning_amd.classify_by_kmers as cbk
    from trio_binning_amd import kmers

    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == 21)
    fa, fb, fq = tmp_path / "la.txt", tmp_path / "lb.txt", tmp_path / "reads21.fa"
    fa.write_text("".join(x + "\n" for x in v["list_a"]))
    fb.write_text("".join(x + "\n" for x in v["list_b"]))
    with open(fq, "w") as fh:
        for i, s in enumerate(v["reads"]):
            fh.write(f">r{i} some comment\n{s}\n")
    oa, ob = orc.table_from_file(str(fa)), orc.table_from_file(str(fb))

    class List:
        def __init__(self, t):
            self.num_kmers, self.k = t.num_kmers, t.k
            self.contents = self

    lists = {str(fa): List(oa), str(fb): List(ob)}
    made = []

    def make(a, b):
        parts = [StubClassifier(d, lambda bases, offs: orc.count_batch(bases, offs, oa, ob), 100 + d) for d in range(n_devices)]
        made.append(kmers.MultiClassifier.from_classifiers(parts))
        return made[-1]

    monkeypatch.setattr(cbk, "_BATCH_BASES", 300)
    monkeypatch.setattr(cbk, "_BATCH_READS", 4)
    monkeypatch.setattr(cbk, "make_classifier", make)
    monkeypatch.setattr(kmers, "create_kmer_hash_set", lambda path: lists[path])
    od = tmp_path / "out"
    od.mkdir()
    with patch("sys.argv", ["classify-by-kmers", str(fq), str(fa), str(fb), "--haplotype-a-out-prefix", str(od / "hapA"),
                            "--haplotype-b-out-prefix", str(od / "hapB"), "--unclassified-out-prefix", str(od / "unclassified")]):
        cbk.main()
    out, _ = capsys.readouterr()
    assert out == v["cli_stdout"]
    assert sorted(os.listdir(od)) == sorted(v["cli_bins"])
    for fn, digest in v["cli_bins"].items():
        assert hashlib.sha256(gzip.open(od / fn, "rb").read()).hexdigest() == digest, fn
    assert made and all(sum(len(p.submitted) for p in m._stubs) >= 30 for m in made)
    assert all(len(p.submitted) > 0 for p in made[0]._stubs)   # every device took batches


def test_device_list_from_environment(built, monkeypatch):
    from trio_binning_amd import kmers

    monkeypatch.setenv("TBK_DEVICES", "2,0,2")
    assert kmers.visible_devices() == [2, 0, 2]
    monkeypatch.delenv("TBK_DEVICE", raising=False)
    assert kmers.default_device() == 2
    monkeypatch.setenv("TBK_DEVICE", "5")
    assert kmers.default_device() == 5
    monkeypatch.delenv("TBK_DEVICES")
    assert kmers.visible_devices() == list(range(__import__("trio_binning_amd")._lib.device_count()))


def test_native_loop_reports_errors_and_handles_empty_input(built, tmp_path):
    """tbk_classify_file over stub rings: a reads file that does not exist is an IOError (the type the reference's
    open() raises), an empty reads file gives three valid empty bins and no TSV line, and a ring that fails a batch
    fails the run instead of writing a partial TSV silently."""
    import io

    from trio_binning_amd import kmers

    multi = kmers.MultiClassifier.from_classifiers([StubClassifier(0, _lens, 1)])
    names = [str(tmp_path / n) for n in ("a.fa.gz", "b.fa.gz", "u.fa.gz")]
    with pytest.raises(IOError):
        multi.classify_file(str(tmp_path / "nope.fa"), 3, 4, names, True, tsv_fd=-1)
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    with open(tmp_path / "tsv", "wb") as fh:
        st = multi.classify_file(str(empty), 3, 4, names, True, tsv_fd=fh.fileno())
    assert st["reads"] == 0 and os.path.getsize(tmp_path / "tsv") == 0
    assert all(gzip.open(n, "rb").read() == b"" for n in names)
    multi.close()

    class Broken(StubClassifier):
        def submit(self, bases, offsets):
            raise RuntimeError("no device")

    broken = kmers.MultiClassifier.from_classifiers([Broken(0, _lens, 1)])
    reads = tmp_path / "r.fa"
    reads.write_text(">r1\nACGTACGTAC\n>r2\nGGGG\n")
    with pytest.raises(Exception):
        broken.classify_file(str(reads), 3, 4, names, False, tsv_fd=-1)
    broken.close()
