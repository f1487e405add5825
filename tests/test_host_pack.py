"""The host packer of the packed transfer format (tbk_pack_bases, host code: no GPU needed) against
a numpy restatement of the device's pack16 (csrc/tbk_device.h): 2-bit codes per base, and the
exceptions for chunks holding a byte outside ACGT or positions past the end of the stream."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def want_packed(bases: np.ndarray):
    total = bases.size
    n_chunks = (total + 15) // 16
    padded = np.zeros(n_chunks * 16, dtype=np.uint8)
    padded[:total] = bases
    code = ((padded >> 1) ^ (padded >> 2)) & 3
    good = np.isin(padded, np.frombuffer(b"ACGT", dtype=np.uint8))
    good[total:] = False
    shifts = (2 * np.arange(16, dtype=np.uint64))
    codes = (code.reshape(n_chunks, 16).astype(np.uint64) << shifts).sum(axis=1).astype(np.uint32)
    mask = ((~good).reshape(n_chunks, 16).astype(np.uint32) << np.arange(16, dtype=np.uint32)).sum(axis=1).astype(np.uint16)
    if total % 16:
        codes[-1] &= np.uint32((1 << (2 * (total % 16))) - 1)
    # where a byte is not ACGT the code bits are free: compare codes under the mask of good bases only
    return codes, mask


def check_pack(kmers, bases):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offs = np.array([0, bases.size], dtype=np.uint64)
    p = kmers.pack_bases(bases, offs, pinned=False)
    codes, mask = want_packed(bases)
    assert p.codes.size == codes.size
    dense = np.zeros(codes.size, dtype=np.uint16)
    assert np.all(np.diff(p.exc_chunk.astype(np.int64)) > 0)         # ordered by chunk, each chunk once
    dense[p.exc_chunk] = p.exc_mask
    assert np.array_equal(dense, mask)
    assert np.all(p.exc_mask != 0)
    good2 = np.repeat(~((mask[:, None] >> np.arange(16)) & 1).astype(bool), 2, axis=1)       # two code bits per base
    keep = (good2.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    assert np.array_equal(p.codes & keep, codes & keep)
    return p


def test_packer_matches_the_device_packing(built):
    from trio_binning_amd import kmers

    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 100, 1000, 4097):
        check_pack(kmers, acgt[rng.integers(0, 4, n)])
    # every byte value, in every position of a chunk
    allb = np.tile(np.arange(256, dtype=np.uint8), 17)
    check_pack(kmers, allb)
    # reads with N runs, lower case, IUPAC; a clean stream has no exceptions but the tail's
    noisy = acgt[rng.integers(0, 4, 300_001)].copy()
    noisy[rng.integers(0, noisy.size, 3000)] = np.frombuffer(b"NnacgtRYKM-*", dtype=np.uint8)[rng.integers(0, 12, 3000)]
    noisy[5000:5400] = ord("N")
    check_pack(kmers, noisy)
    clean = check_pack(kmers, acgt[rng.integers(0, 4, 64_000)])
    assert clean.exc_chunk.size == 0
    assert clean.nbytes == 64_000 // 4 + 16     # a quarter byte per base + the two offsets
    tail = check_pack(kmers, acgt[rng.integers(0, 4, 64_005)])
    assert tail.exc_chunk.tolist() == [4000] and tail.exc_mask.tolist() == [0xFFE0]


def test_packer_threads_and_scalar_path_agree(built):
    """A batch big enough for several packer threads, and the scalar code path (TBK_NO_AVX2=1) in a
    child process: same words, same exceptions."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(9)
    n = 40_000_003
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    bad = rng.integers(0, n, 20_000)
    bases[bad] = ord("N")
    p = check_pack(kmers, bases)
    path = "/tmp/tbk_pack_test_%d.npy" % os.getpid()
    np.save(path, bases)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from trio_binning_amd import kmers;"
            "b = np.load(%r); p = kmers.pack_bases(b, np.array([0, b.size], dtype=np.uint64), pinned=False);"
            "import zlib; print(zlib.crc32(p.codes.tobytes()), zlib.crc32(p.exc_chunk.tobytes()), zlib.crc32(p.exc_mask.tobytes()))" % (ROOT, path))
    try:
        outs = []
        for env in ({"TBK_NO_AVX2": "1", "TBK_HOST_THREADS": "1"}, {"TBK_HOST_THREADS": "5"}):
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-1500:]
            outs.append(r.stdout.split())
    finally:
        os.unlink(path)
    import zlib

    # code bits of not-ACGT bytes are free, and the scalar and AVX2 paths happen to agree on them too
    mine = [str(zlib.crc32(p.codes.tobytes())), str(zlib.crc32(p.exc_chunk.tobytes())), str(zlib.crc32(p.exc_mask.tobytes()))]
    assert outs[0][1:] == outs[1][1:] == mine[1:]
    assert outs[0][0] == outs[1][0] == mine[0]


def test_pack_bases_capacity_protocol(built):
    import ctypes as C

    from trio_binning_amd import _lib
    from trio_binning_amd._lib import lib

    bases = np.frombuffer(b"ACGTNACGTACGTACGTNNNNACGTACGTACGTACGTAC", dtype=np.uint8)
    codes = np.zeros(3, dtype=np.uint32)
    n = C.c_uint64()
    ec, em = np.zeros(1, dtype=np.uint32), np.zeros(1, dtype=np.uint16)
    assert lib.tbk_packed_chunks(bases.size) == 3
    assert lib.tbk_pack_bases(bases.ctypes.data, bases.size, codes.ctypes.data, ec.ctypes.data, em.ctypes.data, 1, C.byref(n)) == _lib.TBK_ERR_NOMEM
    assert n.value == 3
    ec, em = np.zeros(3, dtype=np.uint32), np.zeros(3, dtype=np.uint16)
    assert lib.tbk_pack_bases(bases.ctypes.data, bases.size, codes.ctypes.data, ec.ctypes.data, em.ctypes.data, 3, C.byref(n)) == 0
    assert ec.tolist() == [0, 1, 2] and em.tolist() == [1 << 4, 0b11110, 0xFF80]
    assert lib.tbk_pack_bases(None, 5, codes.ctypes.data, None, None, 0, C.byref(n)) == _lib.TBK_ERR_INVALID


def _fastq_text(rng, n, lo, hi, bad_every=0):
    out = []
    nrng = np.random.default_rng(rng.randrange(1 << 30))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n):
        ln = rng.randrange(lo, hi)
        s = acgt[nrng.integers(0, 4, ln)].tobytes().decode()
        if bad_every and i % bad_every == 0 and ln > 3:
            p = rng.randrange(ln)
            s = s[:p] + rng.choice("Nnacgt-") + s[p + 1:]
        out.append(f"@r{i} c\n{s}\n+\n{'I' * ln}\n")
    return "".join(out)


@pytest.mark.parametrize("kind", ["plain", "gz", "fasta", "inflated_scan"])
def test_reader_emits_the_packed_form_of_its_batches(built, tmp_path, monkeypatch, kind):
    """BatchReader(packing=True): every batch carries its bases in the packed transfer format, identical to
    what tbk_pack_bases makes of the same bases - whether the chunk-parallel scan packed the records as it
    copied them (plain FASTQ; several threads, several windows per batch) or the batch was packed in one go
    (records that came through the sequential machine)."""
    import gzip
    import random

    from trio_binning_amd import kmers, seq

    rng = random.Random(11)
    monkeypatch.setenv("TBK_HOST_THREADS", "5")
    if kind == "fasta":
        text = "".join(f">s{i}\n{''.join(rng.choice('ACGTN') for _ in range(rng.randrange(0, 300)))}\n" for i in range(500))
        path = tmp_path / "r.fa"
        path.write_text(text)
    else:
        # big enough for the scan to cut it into several pieces (4 MiB apiece), with tiny and empty reads and bad bytes
        text = _fastq_text(rng, 3000, 0, 40, bad_every=7) + _fastq_text(rng, 2600, 9000, 21000, bad_every=50) + _fastq_text(rng, 500, 1, 18)
        if kind == "plain":
            path = tmp_path / "r.fq"
            path.write_text(text)
        else:
            path = tmp_path / "r.fq.gz"
            with gzip.open(path, "wt") as fh:
                fh.write(text)
            if kind == "inflated_scan":
                monkeypatch.setenv("TBK_INFLATED_SCAN", "1")
    n_batches = 0
    for limit in (0, 7_000_000, 1_000_003):
        with seq.BatchReader(str(path), packing=True) as r:
            b = seq.Batch()
            while r.next_batch(b, limit, 0):
                n_batches += 1
                bases, off = b.arrays()[0], b.arrays()[1]
                got = b.packed_arrays()
                assert got is not None
                want = kmers.pack_bases(bases, off, pinned=False)
                assert np.array_equal(got[0], want.codes)
                assert sorted(zip(got[1].tolist(), got[2].tolist())) == sorted(zip(want.exc_chunk.tolist(), want.exc_mask.tolist()))
            b.close()
    assert n_batches >= 3
    with seq.BatchReader(str(path)) as r:  # packing off: no packed form
        b = seq.Batch()
        assert r.next_batch(b, 0, 0) and b.packed_pointers() is None
        b.close()


def _mixed_fastq(rng, n, lo, hi, bad_every=0):
    """Regular 4-line records in every shape the borrowed path distinguishes: with and without a header comment,
    bare and named '+' lines, empty sequences, bytes outside ACGT."""
    out = []
    nrng = np.random.default_rng(rng.randrange(1 << 30))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n):
        ln = rng.randrange(lo, hi)
        s = acgt[nrng.integers(0, 4, ln)].tobytes().decode()
        if bad_every and i % bad_every == 0 and ln > 3:
            p = rng.randrange(ln)
            s = s[:p] + rng.choice("Nnacgt-") + s[p + 1:]
        q = "".join(chr(33 + (j * 7 + i) % 40) for j in range(ln)).replace("@", "A").replace("+", "B") if ln < 200 else chr(40 + i % 30) * ln
        name = f"m{i}/{rng.randrange(10**6)}/ccs"
        head = name + (f" np={i} rq=0.99" if rng.random() < 0.4 else "")
        plus = "+" + (name if rng.random() < 0.3 else "")
        out.append(f"@{head}\n{s}\n{plus}\n{q}\n")
        _MIXED_RECORDS.append((name, s, q))
    return "".join(out)


_MIXED_RECORDS = []


@pytest.mark.parametrize("shape", ["long", "short", "mixed"])
def test_borrowed_batches_equal_copied_ones(built, tmp_path, monkeypatch, shape):
    """BatchReader(packing=True, borrowing=True) - what the native loop reads with: batches of a plain FASTQ file leave
    their records in the reader's mapping.  Names, offsets and the packed form equal those of the copied batch of the
    same records, and the bin writer writes the same bytes from the mapping (plain long records: gathered pwritev;
    short records: bin buffers; gzip members) as from the copied arrays - the bytes Read.print would write."""
    import gzip
    import random

    from trio_binning_amd import seq

    rng = random.Random({"long": 5, "short": 6, "mixed": 7}[shape])
    monkeypatch.setenv("TBK_HOST_THREADS", "5")
    del _MIXED_RECORDS[:]
    if shape == "long":     # several scan pieces and several copy threads per batch
        text = _mixed_fastq(rng, 2500, 9000, 21000, bad_every=40)
    elif shape == "short":  # many records per 16-base chunk, empty sequences, chunks spanning several records
        text = _mixed_fastq(rng, 60000, 0, 40, bad_every=7) + _mixed_fastq(rng, 1500, 9000, 21000)
    else:
        text = _mixed_fastq(rng, 3000, 0, 40, bad_every=7) + _mixed_fastq(rng, 2600, 9000, 21000, bad_every=50) + _mixed_fastq(rng, 500, 1, 18)
    path = tmp_path / "r.fq"
    path.write_text(text)
    records = list(_MIXED_RECORDS)  # (name, seq, qual) of every record, for the expected bins

    def expected(bins_all):
        out = {b"A": [], b"B": [], b"U": []}
        k = 0
        for name, s, q in records:
            out[bins_all[k:k + 1]].append(f"@{name}\n{s}\n+\n{q}\n" if q else f">{name}\n{s}\n")
            k += 1
        return {b: "".join(v).encode() for b, v in out.items()}

    for limit in (0, 7_000_000, 1_000_003):
        results = {}
        for borrowing in (False, True):
            for gz in (False, True):
                prefix = [str(tmp_path / f"{x}_{int(borrowing)}_{int(gz)}_{limit}") for x in "abu"]
                w = seq.BinWriter(prefix[0], prefix[1], prefix[2], ".fq", gz)
                meta, bins_all = [], b""
                brng = random.Random(limit)
                with seq.BatchReader(str(path), packing=True, borrowing=borrowing) as r:
                    b = seq.Batch()
                    while r.next_batch(b, limit, 0):
                        assert b.borrowed == borrowing
                        a = b.arrays()
                        got = b.packed_arrays()
                        assert got is not None
                        meta.append((a[1].copy(), a[2].copy(), a[3].copy(), a[5].copy(), a[6].copy(), got[0].copy(),
                                     sorted(zip(got[1].tolist(), got[2].tolist()))))
                        # runs of one bin (neighbours merge into one piece) and single records
                        bins = b"".join(brng.choice([b"A", b"B", b"U"]) * brng.choice([1, 1, 2, 5]) for _ in range(b.n_reads))[:b.n_reads]
                        w.write(b, bins)
                        bins_all += bins
                    b.close()
                    w.close()  # (borrowed batches are valid until the reader is closed)
                data = {}
                for key, name in zip((b"A", b"B", b"U"), w.names):
                    raw = open(name, "rb").read()
                    data[key] = gzip.decompress(raw) if gz else raw
                results[(borrowing, gz)] = (meta, data, bins_all)
        base_meta, base_data, bins_all = results[(False, False)]
        want = expected(bins_all)
        assert base_data == want
        for key, (meta, data, bins) in results.items():
            assert bins == bins_all and data == want, key
            assert len(meta) == len(base_meta)
            for m, bm in zip(meta, base_meta):
                for x, y in zip(m[:6], bm[:6]):
                    assert np.array_equal(x, y), key
                assert m[6] == bm[6], key


def test_borrowed_batch_outlives_its_reader(built, tmp_path, monkeypatch):
    """A borrowed batch keeps the reader's mapping alive (refilling one after tbk_fastx_close let
    MADV_DONTNEED loose on whatever had been mapped there since).  Here: fill a batch from a.fq with borrowing,
    close the reader, allocate over the freed address range, refill the same batch from b.fq - the arrays allocated
    in between keep their contents, the old batch was still readable after the close, and the refilled batch holds
    b.fq's records.  Batch.reads() of a borrowed batch returns the real sequences and qualities (gathered from the
    mapping), Batch.pointers() refuses."""
    import random

    from trio_binning_amd import seq

    monkeypatch.setenv("TBK_HOST_THREADS", "3")
    rng = random.Random(11)
    del _MIXED_RECORDS[:]
    text_a = _mixed_fastq(rng, 1200, 9000, 21000, bad_every=40)  # ~18 MB: many whole pages inside the batch
    recs_a = list(_MIXED_RECORDS)
    del _MIXED_RECORDS[:]
    text_b = _mixed_fastq(rng, 300, 9000, 21000)
    recs_b = list(_MIXED_RECORDS)
    (tmp_path / "a.fq").write_text(text_a)
    (tmp_path / "b.fq").write_text(text_b)

    b = seq.Batch()
    ra = seq.BatchReader(str(tmp_path / "a.fq"), packing=True, borrowing=True)
    assert ra.next_batch(b, 0, 0) == len(recs_a) and b.borrowed
    with pytest.raises(ValueError):
        b.pointers()
    ra.close()
    # the mapping is still there: the records read back after the close
    got = b.reads()
    assert [(r.name, r.seq, r.qual) for r in got] == recs_a
    # memory allocated now must survive the refill (it would land where the mapping was, had that been unmapped)
    live = [np.ones(1 << 21, dtype=np.uint64) for _ in range(6)]
    with seq.BatchReader(str(tmp_path / "b.fq"), packing=True, borrowing=True) as rb:
        assert rb.next_batch(b, 0, 0) == len(recs_b) and b.borrowed
        assert [(r.name, r.seq, r.qual) for r in b.reads()] == recs_b
    assert all(int(x.min()) == 1 and int(x.max()) == 1 for x in live)
    # ... and again after that reader is gone too
    assert [(r.name, r.seq, r.qual) for r in b.reads()] == recs_b
    b.close()


def test_lognormal_lengths_are_deterministic_and_shaped(built):
    """tbk_synth_lognormal_lengths (host code): BASELINE configs[4]'s read lengths - N50 ~ 100 kb, a tail past 1 Mb, a floor of
    short reads; a read's length depends on (seed, index) alone, whatever batch it is drawn in."""
    import ctypes as C

    import numpy as np

    from trio_binning_amd._lib import check, lib

    n = 200_000
    offs = np.zeros(n + 1, dtype=np.uint64)
    check(lib.tbk_synth_lognormal_lengths(0x5EED0002, 0, n, 100_000.0, 0.9, 0.05, 1, 4_000_000, offs.ctypes.data))
    lens = np.diff(offs).astype(np.int64)
    order = np.sort(lens)[::-1]
    n50 = order[np.searchsorted(np.cumsum(order), lens.sum() / 2)]
    assert 95_000 < n50 < 105_000, n50
    assert lens.max() > 1_000_000 and lens.max() <= 4_000_000 and lens.min() >= 1
    assert 0.045 < (lens <= 5000).mean() < 0.06
    assert 60_000 < lens.mean() < 68_000
    part = np.zeros(1001, dtype=np.uint64)
    check(lib.tbk_synth_lognormal_lengths(0x5EED0002, 12_345, 1000, 100_000.0, 0.9, 0.05, 1, 4_000_000, part.ctypes.data))
    assert np.array_equal(np.diff(part).astype(np.int64), lens[12_345:13_345])
    with pytest.raises(ValueError):
        check(lib.tbk_synth_lognormal_lengths(1, 0, 10, 100_000.0, -1.0, 0.05, 1, 4_000_000, part.ctypes.data))
