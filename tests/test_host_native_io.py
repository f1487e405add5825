"""Native I/O either side of the path (tbk_fastx_*, tbk_bin_writer_*, tbk_format_tsv): host
code of the C-ABI library, testable without a GPU.  The reader must yield exactly the
records the reference's readfq yields (golden vectors recorded from the real reference),
the writer exactly the bytes of Read.print, the TSV exactly Python's str(float)."""
import gzip
import io
import os
import random
import struct

import numpy as np
import pytest

from conftest import DATA, ROOT, load_golden


def _native_records(path, max_bases=0, max_reads=0):
    from trio_binning_amd import seq

    out = []
    with seq.BatchReader(path) as r:
        b = seq.Batch()
        while r.next_batch(b, max_bases, max_reads):
            out += [[x.name, x.seq, x.qual] for x in b.reads()]
        b.close()
    return out


@pytest.mark.parametrize("limits", [(0, 0), (1, 0), (0, 1), (0, 3), (7, 2)])
def test_reader_quirk_vectors(built, tmp_path, limits):
    g = load_golden("readfq_vectors.json")
    for name, case in g.items():
        data = bytes.fromhex(case["bytes_hex"]) if name == "crlf_file" else case["text"].encode()
        p = tmp_path / "x.fastx"
        p.write_bytes(data)
        assert _native_records(str(p), *limits) == case["records"], name


def test_reader_fuzz_corpus(built, tmp_path):
    """400 random byte strings parsed by the real reference (tests/golden/readfq_fuzz.json)."""
    from trio_binning_amd import seq

    p = tmp_path / "fuzz.fx"
    gz = tmp_path / "fuzz.fx.gz"
    for i, case in enumerate(load_golden("readfq_fuzz.json")):
        data = bytes.fromhex(case["bytes_hex"])
        p.write_bytes(data)
        assert _native_records(str(p)) == case["records"], (i, data)
        assert _native_records(str(p), 0, 1) == case["records"], (i, data)
        # our Python mirror agrees too
        assert [[r.name, r.seq, r.qual] for r in seq.open_fastx_read(str(p))] == case["records"], (i, data)
        if i % 8 == 0:
            with gzip.open(gz, "wb") as fh:
                fh.write(data)
            assert _native_records(str(gz)) == case["records"], (i, data)


def test_reader_equals_python_mirror_on_toy_files(built, tmp_path):
    from trio_binning_amd import seq

    for name in ("test.fa", "test.fastq", "test.ccs.fastq.gz"):
        path = os.path.join(DATA, name)
        want = [[r.name, r.seq, r.qual] for r in seq.open_fastx_read(path)]
        for limits in ((0, 0), (10000, 0), (0, 1)):
            assert _native_records(path, *limits) == want
    # multi-member gzip, CR-only newlines, a 3 MB single-line record, long multi-line records
    rng = random.Random(5)
    big = "".join(rng.choice("ACGT") for _ in range(3_000_000))
    text = ">big one\n" + big + "\n>wrapped\n" + "\n".join(big[i:i + 70] for i in range(0, 50_000, 70)) + "\n"
    text += "@q1 x\n" + big[:100_000] + "\n+\n" + "I" * 100_000 + "\n"
    plain = tmp_path / "big.fa"
    plain.write_text(text)
    want = [[r.name, r.seq, r.qual] for r in seq.open_fastx_read(str(plain))]
    assert len(want) == 3 and len(want[0][1]) == 3_000_000 and want[2][2] == "I" * 100_000
    assert _native_records(str(plain)) == want
    multi = tmp_path / "multi.fa.gz"
    with open(multi, "wb") as fh:
        third = len(text) // 3
        for part in (text[:third], text[third:2 * third], text[2 * third:]):
            fh.write(gzip.compress(part.encode()))
    assert _native_records(str(multi)) == want
    cr = tmp_path / "cr.fa"
    cr.write_bytes(b">a b\rACGT\rGG\r>c\rTT")
    assert _native_records(str(cr)) == [[r.name, r.seq, r.qual] for r in seq.open_fastx_read(str(cr))]


def _bgzf(data: bytes, block=60000, eof_marker=True) -> bytes:
    """bgzip's container: gzip members of <= 64 KiB, each carrying its size in a BC extra subfield."""
    import zlib

    out = bytearray()
    pieces = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for piece in pieces:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        bsize = 18 + len(body) + 8
        out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)


def test_reader_inflates_bgzf_blocks_side_by_side(built, tmp_path):
    """bgzip'ed input (independent <= 64 KiB members) goes through the block-parallel inflate; the
    records must be those of the plain file.  Also: BGZF blocks followed by an ordinary gzip member
    (the sequential path takes over), an ordinary member first, and damaged files."""
    from trio_binning_amd import _lib

    rng = random.Random(11)
    seqs = ["".join(rng.choice("ACGTN") for _ in range(rng.randint(1, 3000))) for _ in range(3000)]
    text = "".join("@r%d c\n%s\n+\n%s\n" % (i, q, "I" * len(q)) for i, q in enumerate(seqs)).encode()
    plain = tmp_path / "p.fastq"
    plain.write_bytes(text)
    want = _native_records(str(plain))
    assert len(want) == 3000
    cases = {
        "bgzf.fastq.gz": _bgzf(text),
        "bgzf_small_blocks.fastq.gz": _bgzf(text, block=777),
        "bgzf_no_eof.fastq.gz": _bgzf(text, eof_marker=False),
        "bgzf_then_gzip.fastq.gz": _bgzf(text[:100_000], eof_marker=False) + gzip.compress(text[100_000:]),
        "gzip_then_bgzf.fastq.gz": gzip.compress(text[:100_000]) + _bgzf(text[100_000:]),
        # zero padding between members, which Python's GzipFile skips: between BGZF blocks, before an
        # ordinary member that follows them, and at the end of the file
        "bgzf_padded.fastq.gz": _bgzf(text[:200_000], eof_marker=False) + b"\0" * 11 + _bgzf(text[200_000:]) + b"\0" * 5,
        "bgzf_pad_gzip.fastq.gz": _bgzf(text[:100_000], eof_marker=False) + b"\0" * 3 + gzip.compress(text[100_000:]),
    }
    for name, blob in cases.items():
        f = tmp_path / name
        f.write_bytes(blob)
        assert gzip.decompress(blob) == text, name  # the container is valid gzip
        assert _native_records(str(f)) == want, name
        assert _native_records(str(f), max_bases=50_000) == want, name
    good = _bgzf(text)
    for name, blob in (("cut.fastq.gz", good[: len(good) // 2]), ("flip.fastq.gz", good[:5000] + bytes([good[5000] ^ 0x55]) + good[5001:])):
        f = tmp_path / name
        f.write_bytes(blob)
        with pytest.raises((_lib.TbkError, ValueError, IOError, OSError)):
            _native_records(str(f))


@pytest.mark.parametrize("guessing", [False, True])
def test_own_inflate_equals_zlib(built, tmp_path, monkeypatch, guessing):
    """Ordinary gzip streams go through the library's own DEFLATE decoder; TBK_INFLATE=zlib keeps
    zlib's.  Same records from both on stored / fast / default / best compression, fixed-Huffman
    blocks, header extras (FEXTRA, FNAME, FCOMMENT), several members with zero padding between them,
    highly repetitive and single-symbol data; damaged streams are refused, never mis-decoded.
    guessing: the same through the several-thread decoder (LineSource::pinflate_loop), made to cut
    even these small files into 4 KiB spans."""
    import zlib

    if guessing:
        monkeypatch.setenv("TBK_PINFLATE_MIN", "0")
        monkeypatch.setenv("TBK_PINFLATE_SPAN", "4096")
    else:
        monkeypatch.setenv("TBK_PINFLATE", "0")

    from trio_binning_amd import _lib

    rng = random.Random(3)

    def fastq(n, maxlen, alphabet="ACGT", qualalpha="I"):
        parts = []
        for i in range(n):
            L = rng.randint(1, maxlen)
            parts.append("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice(alphabet) for _ in range(L)), "".join(rng.choice(qualalpha) for _ in range(L))))
        return "".join(parts).encode()

    cases = {
        "dna_const": fastq(400, 3000),
        "dna_qual": fastq(300, 3000, qualalpha="".join(chr(33 + i) for i in range(60))),
        "repeat": ("@r\n" + "ACGTACGTAA" * 30000 + "\n+\n" + "I" * 300000 + "\n").encode() * 2,
        "one_symbol": fastq(300, 2000, alphabet="A"),
        "tiny": b"@a\nACGT\n+\nIIII\n",
        "empty": b"",
    }

    def both(path):
        monkeypatch.setenv("TBK_INFLATE", "zlib")
        a = _native_records(str(path), max_bases=1 << 18)
        monkeypatch.setenv("TBK_INFLATE", "own")
        b = _native_records(str(path), max_bases=1 << 18)
        assert a == b, path
        return b

    for name, data in cases.items():
        want = None
        for level in (0, 1, 6, 9):
            f = tmp_path / f"{name}_{level}.fastq.gz"
            f.write_bytes(gzip.compress(data, level))
            got = both(f)
            want = got if want is None else want
            assert got == want
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
        body = co.compress(data) + co.flush()
        member = (b"\x1f\x8b\x08\x1c" + b"\0" * 6 + struct.pack("<H", 5) + b"hello" + b"name.fq\0" + b"a comment\0" + body
                  + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF))
        f = tmp_path / f"{name}_fixed_multi.fastq.gz"
        f.write_bytes(member + b"\0" * 7 + gzip.compress(b"", 6))
        assert both(f) == want, name
    good = gzip.compress(cases["dna_qual"], 6)
    monkeypatch.setenv("TBK_INFLATE", "own")
    for trial in range(30):
        blob = bytearray(good)
        if trial % 3 == 0:
            blob = blob[: rng.randint(20, len(blob) - 1)]
        elif trial % 3 == 1:
            for _ in range(3):
                blob[rng.randint(12, len(blob) - 1)] ^= 1 << rng.randint(0, 7)
        else:
            blob[-5] ^= 0x40  # the stored CRC
        f = tmp_path / "bad.fastq.gz"
        f.write_bytes(bytes(blob))
        with pytest.raises((_lib.TbkError, ValueError, IOError, OSError)):
            _native_records(str(f))
    # a dynamic block whose literal/length code leaves code space unused (two 2-bit codes: 'A' and
    # end-of-block) is refused by zlib ("invalid literal/lengths set") and by the library's decoder
    bits = []

    def put(v, n):
        bits.extend((v >> i) & 1 for i in range(n))

    put(1, 1); put(2, 2)                    # BFINAL, dynamic Huffman
    put(0, 5); put(0, 5); put(14, 4)        # HLIT = 257, HDIST = 1, HCLEN = 18
    cl_order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1]
    cl_len = {0: 1, 2: 2, 18: 2}            # a complete code-length code: 0 -> '0', 2 -> '10', 18 -> '11'
    for sym in cl_order:
        put(cl_len.get(sym, 0), 3)

    def cl(sym):                            # Huffman codes go out most significant bit first
        code, n = {0: (0, 1), 2: (2, 2), 18: (3, 2)}[sym]
        bits.extend((code >> (n - 1 - i)) & 1 for i in range(n))

    cl(18); put(65 - 11, 7)                 # 65 zeros
    cl(2)                                   # 'A': 2 bits
    cl(18); put(138 - 11, 7); cl(18); put(52 - 11, 7)   # 190 zeros
    cl(2)                                   # end-of-block: 2 bits
    cl(0)                                   # no distance code
    put(0, 2); put(1, 2)                    # 'A', end-of-block
    raw = bytes(sum(b << i for i, b in enumerate(bits[j:j + 8])) for j in range(0, len(bits), 8))
    with pytest.raises(zlib.error):
        zlib.decompress(raw, -15)
    f = tmp_path / "incomplete.fastq.gz"
    f.write_bytes(b"\x1f\x8b\x08\x00" + b"\0" * 6 + raw + struct.pack("<II", zlib.crc32(b"A"), 1))
    with pytest.raises((_lib.TbkError, ValueError, IOError, OSError)):
        _native_records(str(f))
    assert "incomplete" in _lib.last_error() or "inflate" in _lib.last_error()
    # arbitrary bytes behind a gzip header, spliced streams, trailing garbage: refused, and (this test
    # also runs under ASan + UBSan) without reading or writing out of bounds
    for trial in range(120):
        kind = trial % 4
        if kind == 0:
            blob = b"\x1f\x8b\x08\x00" + b"\0" * 6 + bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 3000)))
        elif kind == 1:
            blob = good[:10] + bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 2000))) + good[rng.randint(10, len(good) - 1):]
        elif kind == 2:
            b = bytearray(good)
            for _ in range(rng.randint(2, 20)):
                b[rng.randint(10, len(b) - 9)] = rng.getrandbits(8)
            blob = bytes(b)
        else:
            blob = good + bytes(rng.getrandbits(8) | 1 for _ in range(rng.randint(1, 50)))
        f = tmp_path / "garbage.fastq.gz"
        f.write_bytes(blob)
        try:
            _native_records(str(f))
            decoded = True
        except (_lib.TbkError, ValueError, IOError, OSError):
            decoded = False
        assert not decoded or kind == 2  # a few random byte changes can leave a valid stream (CRC collisions aside: text identical)


def test_reader_batch_layout(built, tmp_path):
    """bases lie back to back, offsets are cumulative, limits cut after whole records."""
    from trio_binning_amd import seq

    p = tmp_path / "r.fq"
    p.write_text("".join(f"@r{i}\n{'ACGT' * (i + 1)}\n+\n{'I' * 4 * (i + 1)}\n" for i in range(10)))
    with seq.BatchReader(str(p)) as r:
        b = seq.Batch()
        sizes = []
        while r.next_batch(b, 30, 0):
            bases, boff, names, noff, quals, qoff, hq = b.arrays()
            assert boff[0] == 0 and len(boff) == b.n_reads + 1 and bases.size == boff[-1]
            assert bytes(bases) == b"".join(x.seq.encode() for x in b.reads())
            assert hq.tolist() == [1] * b.n_reads
            sizes.append(int(boff[-1]))
    assert sum(sizes) == sum(4 * (i + 1) for i in range(10)) and all(s >= 30 for s in sizes[:-1])
    with pytest.raises(IOError):
        seq.BatchReader(str(tmp_path / "missing.fq"))


@pytest.mark.parametrize("gz", [True, False])
def test_writer_bytes_equal_read_print(built, tmp_path, gz):
    from trio_binning_amd import seq

    rng = random.Random(9)
    recs = []
    for i in range(300):
        n = rng.randrange(0, 400)
        s = "".join(rng.choice("ACGT") for _ in range(n))
        kind = rng.random()
        q = None if kind < 0.4 else ("" if kind < 0.5 else "".join(rng.choice("!#5I~") for _ in range(n)))
        recs.append(seq.Read(f"read{i}/x", s, q))
    src = tmp_path / "in.fq"
    with open(src, "w") as fh:
        for r in recs:
            r.print(file=fh)
    bins_all = "".join(rng.choice("ABU") for _ in recs)
    # expected: the Python mirror's writer
    outs = seq.open_outfiles(str(tmp_path / "pa"), str(tmp_path / "pb"), str(tmp_path / "pu"), ".fq", gz)
    parsed = list(seq.open_fastx_read(str(src)))
    for r, b in zip(parsed, bins_all):
        r.print(file=outs["ABU".index(b)])
    for fh in outs:
        fh.close()
    w = seq.BinWriter(str(tmp_path / "na"), str(tmp_path / "nb"), str(tmp_path / "nu"), ".fq", gz, threads=3)
    got_n = 0
    with seq.BatchReader(str(src)) as r:
        b = seq.Batch()
        while r.next_batch(b, 5000, 0):
            w.write(b, bins_all[got_n:got_n + b.n_reads].encode())
            got_n += b.n_reads
    w.close()
    assert got_n == len(parsed)
    opener = (lambda p: gzip.open(p, "rb")) if gz else (lambda p: open(p, "rb"))
    for py_name, nat_name in zip(seq.output_names(str(tmp_path / "pa"), str(tmp_path / "pb"), str(tmp_path / "pu"), ".fq", gz), w.names):
        assert opener(nat_name).read() == opener(py_name).read()


def test_writer_empty_bins_are_valid_files(built, tmp_path):
    from trio_binning_amd import seq

    w = seq.BinWriter(str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "u"), ".fa", True)
    w.close()
    for n in w.names:
        assert gzip.open(n, "rb").read() == b""
    w = seq.BinWriter(str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "u"), ".fa", False)
    w.close()
    assert all(os.path.getsize(n) == 0 for n in w.names)


def test_long_records_go_straight_to_the_files(built, tmp_path):
    """Plain output of long records is gathered from the batch's arrays straight into the files (pwritev from
    several threads, no bin buffer): same bytes as the Python mirror's writer, over several batches, FASTQ and
    FASTA records mixed, a record without a name among them."""
    from trio_binning_amd import seq

    rng = random.Random(21)
    nrng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(90):
        n = rng.randrange(6000, 30000)
        s = acgt[nrng.integers(0, 4, n)].tobytes().decode()
        q = None if i % 7 == 3 else "".join(chr(33 + int(x)) for x in nrng.integers(0, 60, n))
        recs.append(seq.Read("" if i == 11 else f"long{i}/x", s, q))
    src = tmp_path / "in.fq"
    with open(src, "w") as fh:
        for r in recs:
            r.print(file=fh)
    bins_all = "".join(rng.choice("ABU") for _ in recs)
    outs = seq.open_outfiles(str(tmp_path / "pa"), str(tmp_path / "pb"), str(tmp_path / "pu"), ".fq", False)
    parsed = list(seq.open_fastx_read(str(src)))
    for r, b in zip(parsed, bins_all):
        r.print(file=outs["ABU".index(b)])
    for fh in outs:
        fh.close()
    w = seq.BinWriter(str(tmp_path / "na"), str(tmp_path / "nb"), str(tmp_path / "nu"), ".fq", False, threads=4)
    got_n = 0
    with seq.BatchReader(str(src)) as r:
        b = seq.Batch()
        while r.next_batch(b, 400_000, 0):
            w.write(b, bins_all[got_n:got_n + b.n_reads].encode())
            got_n += b.n_reads
    w.close()
    assert got_n == len(parsed)
    for py_name, nat_name in zip(seq.output_names(str(tmp_path / "pa"), str(tmp_path / "pb"), str(tmp_path / "pu"), ".fq", False), w.names):
        assert open(nat_name, "rb").read() == open(py_name, "rb").read()


def test_plain_bin_streams_into_a_fifo(built, tmp_path):
    """A plain bin whose target cannot seek (a FIFO, /dev/stdout through a pipe) is written with
    sequential write(), as the reference's open(name, 'w') handle would; regular files keep pwrite."""
    import threading

    from trio_binning_amd import seq

    src = tmp_path / "in.fq"
    recs = [seq.Read(f"r{i}", "ACGT" * (5 + i % 7), "I" * (4 * (5 + i % 7))) for i in range(200)]
    with open(src, "w") as fh:
        for r in recs:
            r.print(file=fh)
    os.mkfifo(tmp_path / "pa.fq")
    got = {}

    def drain():
        with open(tmp_path / "pa.fq", "rb") as fh:
            got["a"] = fh.read()

    t = threading.Thread(target=drain)
    t.start()
    w = seq.BinWriter(str(tmp_path / "pa"), str(tmp_path / "pb"), str(tmp_path / "pu"), ".fq", False, threads=3)
    bins = ("AB" * len(recs))[:len(recs)]
    n = 0
    with seq.BatchReader(str(src)) as r:
        b = seq.Batch()
        while r.next_batch(b, 50, 0):
            w.write(b, bins[n:n + b.n_reads].encode())
            n += b.n_reads
    w.close()
    t.join(30)
    assert not t.is_alive()
    import io

    want = io.StringIO()
    for r, c in zip(recs, bins):
        if c == "A":
            r.print(file=want)
    assert got["a"].decode() == want.getvalue()
    want_b = io.StringIO()
    for r, c in zip(recs, bins):
        if c == "B":
            r.print(file=want_b)
    assert open(tmp_path / "pb.fq").read() == want_b.getvalue()


def test_float_format_is_python_str(built):
    import ctypes as C

    from trio_binning_amd._lib import lib

    rng = random.Random(3)
    vals = [0.0, 1.0, 4.0, 1.3333333333333333, 2.6666666666666665, 1e15, 1e16, 9999999999999998.0, 1.5e16, 2e18,
            123456789.0, 0.1, 1e-4, 9.999e-5, 1e-5, 5e-324, 1.7976931348623157e308, 3.0000000000000004, 100.0, 1e22, 123456.789]
    vals += [rng.randrange(0, 2**31) * (rng.randrange(1, 10**9) / rng.randrange(1, 10**9)) for _ in range(3000)]
    vals += [struct.unpack("<d", struct.pack("<Q", rng.getrandbits(62)))[0] for _ in range(3000)]
    buf = C.create_string_buffer(64)
    for v in vals:
        n = lib.tbk_format_float(v, buf, 64)
        assert buf.raw[:n].decode() == str(v), v


def test_format_tsv(built, tmp_path):
    from trio_binning_amd import kmers, seq

    p = tmp_path / "r.fa"
    p.write_text(">a desc\nACGT\n>b\tx\nGG\n>\nTT\n")
    with seq.BatchReader(str(p)) as r:
        b = seq.Batch()
        assert r.next_batch(b) == 3
        counts = np.array([[4, 1], [0, 2], [0, 0]], dtype=np.int32)
        sa, sb, bins = kmers.score_and_bin(counts, 4, 3)
        assert seq.format_tsv(b, bins, sa, sb) == "a\tA\t4.0\t1.3333333333333333\nb\tx\tB\t0.0\t2.6666666666666665\n\tU\t0.0\t0.0\n"


def test_list_parser_regular_and_general_paths(built, orc, tmp_path):
    """tbk_list_parse_file (host code): the all-threads fast path for regular lists and the
    general getline-rule parser give the keys the oracle packs from the same file."""
    from trio_binning_amd import kmers

    rng = random.Random(12)

    from test_oracle_golden import SHORT_LINE_LISTS, getline_keys

    def want(text, k):
        kk, keys = getline_keys(text.encode())
        assert kk == k
        return keys

    k = 21
    lines = ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(200_000)]
    cases = {
        "regular": "".join(x + "\n" for x in lines),
        "regular_no_final_newline": "\n".join(lines),
        "with_N_and_lowercase": "".join(x + "\n" for x in lines[:1000]) + "ACGTNNNNacgtACGTACGTA\n" + lines[5] + "\n",
        "one_long_line": "".join(x + "\n" for x in lines[:70000]) + lines[3] + "GGGG\n" + "".join(x + "\n" for x in lines[:70000]),
        "crlf": "".join(x + "\r\n" for x in lines[:70000]),
        "single_line": lines[0] + "\n",
        "tiny_k": "ACG\nTTT\nGGA\nCCC\n",
        "blank_last_line_after_regular": "".join(x + "\n" for x in lines[:70000]) + "\n",
    }
    cases.update({name: data.decode() for name, data in SHORT_LINE_LISTS.items()})
    for name, text in cases.items():
        p = tmp_path / (name + ".txt")
        p.write_bytes(text.encode())
        keys, kk = kmers.parse_kmer_list(str(p))
        first = text.split("\n")[0]
        assert kk == len(first.encode()) + (1 if "\n" in text else 0) - 1, name
        assert keys.tolist() == want(text, kk), name
        t = orc.table_from_file(str(p))
        assert (t.k, t.num_kmers) == (kk, keys.size), name
        if keys.size < 1000:
            assert t.keys().tolist() == sorted(keys.tolist()), name
    for bad in ("", "A" * 33 + "\n", "\n"):
        p = tmp_path / "bad.txt"
        p.write_text(bad)
        with pytest.raises(ValueError):
            kmers.parse_kmer_list(str(p))
    with pytest.raises(IOError):
        kmers.parse_kmer_list(str(tmp_path / "missing.txt"))


@pytest.mark.parametrize("gz", [False, True])
def test_chunk_parallel_scan_equals_the_sequential_machine(built, tmp_path, monkeypatch, gz):
    """FASTQ is read by the chunk-parallel scan while its records are regular - a plain file from its
    mapping, gzip input (gz) from the inflated text at hand, window by window; the sequential
    state machine (TBK_FASTQ_SCAN=0) is the yardstick.  A file big enough for several pieces per
    window, several batch limits, and files where a record stops being regular somewhere in the
    middle (CRLF, wrapped sequence, '>' record, blank line, short or long quality, '+' or '@' first in
    a sequence line, no final newline): the scan must hand over at that record, not a byte early or late."""
    rng = random.Random(77)

    def rec(i, lo=200, hi=9000, qual=None):
        n = rng.randint(lo, hi)
        s = "".join(rng.choices("ACGTN", k=n))
        q = qual if qual is not None else "".join(rng.choices("!#5?@IJ+>", k=n))
        return "@r%d extra words %d\n%s\n+%s\n%s\n" % (i, i, s, "" if i % 3 else "r%d" % i, q)

    ext = ".fastq.gz" if gz else ".fastq"
    if gz:
        monkeypatch.setenv("TBK_INFLATED_SCAN", "1")   # the scan over inflated text is opt-in

    def write(path, text):
        path.write_bytes(gzip.compress(text.encode(), 1) if gz else text.encode())

    def both(path, *limits):
        monkeypatch.setenv("TBK_FASTQ_SCAN", "0")
        want = _native_records(str(path), *limits)
        monkeypatch.setenv("TBK_FASTQ_SCAN", "1")
        assert _native_records(str(path), *limits) == want, (str(path), limits)
        return want

    monkeypatch.setenv("TBK_HOST_THREADS", "6")   # read once per process; harmless if another test got there first
    n_big = 3000 if os.environ.get("TBK_TEST_LIGHT") else 9000   # the sanitized re-run (test_host_sanitizers.py) takes the smaller file
    big = "".join(rec(i) for i in range(n_big))    # ~80 MB: windows of several 4 MB pieces
    p = tmp_path / ("big" + ext)
    write(p, big)
    recs = both(p)
    assert len(recs) == n_big and recs[0][0] == "r0" and recs[-1][0] == "r%d" % (n_big - 1)
    for limits in ((3_000_000, 0), (0, 1000), (10_000_000, 777), (1, 0)):
        assert both(p, *limits) == recs
    # zero-length reads, '@' and '+' leading quality lines, a header that is just '@'
    odd = "@\n\n+\n\n" + "@a b\nACGT\n+\n@@@@\n" + "@c\nAC\n+c\n++\n" + "".join(rec(i, 1, 50) for i in range(2000))
    q = tmp_path / ("odd" + ext)
    write(q, odd)
    assert len(both(q)) == 2003 and both(q, 0, 1) == both(q)
    # irregular records in the middle: everything before them by the scan, everything after by the machine
    head = "".join(rec(i, 50, 4000) for i in range(3000))
    tail = "".join(rec(5000 + i, 50, 4000) for i in range(500))
    breakers = {
        "crlf": "@x\r\nACGT\r\n+\r\nIIII\r\n",
        "cr_in_seq": "@x\nAC\rGT\n+\nIIIII\n",
        "wrapped": "@x\nACGT\nACGT\n+\nIIII\nIIII\n",
        "fasta": ">x\nACGTACGT\n",
        "blank": "\n",
        "short_qual": "@x\nACGTACGT\n+\nIII\n",
        "long_qual": "@x\nACGT\n+\nIIIIIIII\n",
        "plus_in_seq": "@x\n+ACGT\n+\nIIIII\n",
        "at_in_seq": "@x\n@ACGT\n+\nIIIII\n",
        "junk": "hello world\n",
    }
    for name, mid in breakers.items():
        f = tmp_path / (name + ext)
        write(f, head + mid + tail)
        got = both(f)
        assert len(got) >= 3000 and got[:3000] == [[x[0], x[1], x[2]] for x in both(f)[:3000]], name
        both(f, 500_000, 0)
    for name, text in (("no_final_newline", head + "@z\nACGT\n+\nIIII"), ("truncated", head + "@z\nACGTACGT\n+\nII"), ("only_header", head + "@z")):
        f = tmp_path / (name + ext)
        write(f, text)
        both(f)
        both(f, 0, 64)


def _gzip_member(data: bytes) -> bytes:
    import ctypes as C

    from trio_binning_amd._lib import lib

    need = C.c_size_t(0)
    assert lib.tbk_gzip_member(data, len(data), None, 0, C.byref(need)) == -6 and need.value >= 18
    out = C.create_string_buffer(need.value)
    assert lib.tbk_gzip_member(data, len(data), out, need.value, C.byref(need)) == 0
    return out.raw[:need.value]


def test_own_gzip_members_are_read_by_zlib_and_by_the_reader(built, tmp_path):
    """tbk_deflate.cpp: dynamic-Huffman members, runs as distance-1 matches.  Any inflater must give the
    bytes back (zlib here, and the library's own inflater through the reader), CRC and size included;
    covered: empty input, one symbol, all 256 values, skewed counts that need the 15-bit length limit,
    blocks with and without line ends, runs of every length, and FASTQ text with noisy / constant /
    mostly-'~' qualities, where the result must not be larger than zlib's Z_HUFFMAN_ONLY resp. Z_RLE."""
    import zlib

    from trio_binning_amd import seq

    rng = np.random.default_rng(5)
    skew = np.concatenate([np.full(1 << k, k, dtype=np.uint8) for k in range(20)])  # Fibonacci-deep tree without a limit
    rng.shuffle(skew)
    cases = [b"", b"A", b"\n", b"AA", b"AB" * 5, bytes(range(256)) * 3, rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),
             b"A" * 200_000 + b"z", skew.tobytes(), b"\n" * 70_000, b"ACGT" * 5000 + b"\n" + b"I" * 20_000 + b"\n",
             b"".join(bytes([65 + r % 7]) * r for r in range(1, 700)),  # runs of every length: 258-byte matches and their remainders
             np.repeat(rng.choice(np.frombuffer(b"#-7<FI", dtype=np.uint8), 40_000), rng.integers(1, 12, 40_000)).tobytes()]
    for data in cases:
        z = _gzip_member(data)
        assert zlib.decompress(z, 31) == data and gzip.decompress(z) == data, len(data)
    n = 1 << 20
    bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes()
    quals = (33 + np.clip(rng.normal(30, 8, n), 0, 60).astype(np.uint8)).tobytes()
    tilde = np.full(n, ord("~"), dtype=np.uint8)
    dips = rng.random(n) < 0.08
    tilde[dips] = 33 + rng.integers(5, 60, int(dips.sum()))
    for read_len, quals, strategy in ((150, quals, zlib.Z_HUFFMAN_ONLY), (15_000, quals, zlib.Z_HUFFMAN_ONLY),
                                      (150, b"I" * n, zlib.Z_RLE), (15_000, tilde.tobytes(), zlib.Z_RLE)):
        text = b"".join(b"@r%d\n" % i + bases[i:i + read_len] + b"\n+\n" + quals[i:i + read_len] + b"\n" for i in range(0, n, read_len))
        z = _gzip_member(text)
        assert zlib.decompress(z, 31) == text
        co = zlib.compressobj(1, zlib.DEFLATED, 31, 8, strategy)
        assert len(z) <= len(co.compress(text) + co.flush()) * 1.01, read_len
        # two members back to back are one gzip file; the reader (own inflater) parses it
        path = tmp_path / f"r{read_len}.fq.gz"
        path.write_bytes(z + z)
        got = sum(1 for _ in seq.open_fastx_read(str(path)))
        want = 2 * len(range(0, n, read_len))
        with seq.BatchReader(str(path)) as r:
            b, total = seq.Batch(), 0
            while r.next_batch(b, 1 << 22, 0):
                total += b.n_reads
        assert got == want == total


def test_writer_gzip_encoders_agree(built, tmp_path):
    """The same batch through the library's own encoder and through zlib (TBK_GZIP_ENCODER=zlib, read
    once per process, so in child interpreters): same decompressed files."""
    import hashlib
    import subprocess
    import sys

    rng = random.Random(4)
    src = tmp_path / "in.fq"
    with open(src, "w") as fh:
        for i in range(400):
            n = rng.randrange(1, 3000)
            fh.write("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(n)), "".join(rng.choice("!#5I~+,-") for _ in range(n))))
    prog = (
        "import sys, gzip, hashlib\n"
        "from trio_binning_amd import seq\n"
        "w = seq.BinWriter(sys.argv[2] + 'a', sys.argv[2] + 'b', sys.argv[2] + 'u', '.fq', True, threads=2)\n"
        "with seq.BatchReader(sys.argv[1]) as r:\n"
        "    b = seq.Batch()\n"
        "    i = 0\n"
        "    while r.next_batch(b, 100000, 0):\n"
        "        w.write(b, bytes('ABU'[(i + j) % 3].encode()[0] for j in range(b.n_reads)))\n"
        "        i += b.n_reads\n"
        "w.close()\n"
        "print(' '.join(hashlib.sha256(gzip.open(n, 'rb').read()).hexdigest() for n in w.names))\n"
    )
    digests = []
    for enc in ("own", "zlib"):
        env = dict(os.environ, TBK_GZIP_ENCODER=enc)
        r = subprocess.run([sys.executable, "-c", prog, str(src), str(tmp_path / enc)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(r.stdout.split())
    assert digests[0] == digests[1] and len(set(digests[0])) == 3


def test_guessing_inflate_equals_the_sequential_decoder(built, tmp_path, monkeypatch, capfd):
    """A gzip stream large enough for the several-thread decoder at realistic span sizes: single member
    (zlib levels 1, 6, 9: blocks of different sizes, matches reaching back across chunk borders),
    several members of uneven size (one ends in every other round), a member of stored blocks (no
    dynamic block to guess at) in front of a compressed one.  Same records, in the same batches, as
    with TBK_PINFLATE=0; a flipped bit deep in the file, a cut-off file and a wrong CRC are refused."""
    import zlib

    from trio_binning_amd import _lib, seq

    rng = np.random.default_rng(8)
    n, length = 1500, 9000
    bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (n, length))
    quals = (33 + np.clip(rng.normal(30, 8, (n, length)), 0, 60)).astype(np.uint8)
    parts = []
    for i in range(n):
        if i % 3 == 1:
            bases[i, : length // 2] = bases[i - 1, length // 4: length // 4 + length // 2]  # something for LZ77 to find
        parts.append(b"@read%d/ccs np=%d\n" % (i, i % 17) + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
    text = b"".join(parts)

    def member(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
        return co.compress(data) + co.flush()

    third = len(text) // 3
    files = {
        "l1": member(text, 1), "l6": member(text, 6), "l9": member(text, 9),
        "members": member(text[:third], 6) + member(text[third:third + 100], 9) + member(text[third + 100:], 1),
        "stored_first": member(text[:third], 0) + b"\0" * 5 + member(text[third:], 6),
        "fixed_blocks": member(text[:third], 6, zlib.Z_FIXED) + member(text[third:], 6),
    }

    def read_all(path):
        sizes, digest = [], __import__("hashlib").sha256()
        with seq.BatchReader(str(path)) as r:
            b = seq.Batch()
            while r.next_batch(b, 3 << 20, 0):
                sizes.append(b.n_reads)
                for a in b.arrays():
                    digest.update(a.tobytes())
        return sizes, digest.hexdigest()

    monkeypatch.setenv("TBK_PINFLATE_MIN", "0")
    monkeypatch.setenv("TBK_INFLATED_SCAN", "1")   # the guessed chunks also feed the window-by-window record scan
    want = None
    for name, blob in files.items():
        f = tmp_path / f"{name}.fastq.gz"
        f.write_bytes(blob)
        monkeypatch.setenv("TBK_PINFLATE", "0")
        monkeypatch.setenv("TBK_FASTQ_SCAN", "0")   # the yardstick: one inflating thread, the sequential record machine
        ref = read_all(f)
        monkeypatch.delenv("TBK_FASTQ_SCAN")
        assert sum(ref[0]) == n
        want = want or ref
        assert ref == want, name
        monkeypatch.delenv("TBK_PINFLATE")
        for span in (300_000, 70_000):
            monkeypatch.setenv("TBK_PINFLATE_SPAN", str(span))
            assert read_all(f) == want, (name, span)
    # the guesses do hold on such a stream: most rounds keep every chunk they decoded
    monkeypatch.setenv("TBK_PINFLATE_SPAN", "200000")
    monkeypatch.setenv("TBK_PINFLATE_TIMING", "1")
    capfd.readouterr()
    assert read_all(tmp_path / "l6.fastq.gz") == want
    monkeypatch.delenv("TBK_PINFLATE_TIMING")
    import re
    rounds = [(int(g), int(k)) for g, k in re.findall(r"(\d+) guesses, (\d+) chunks kept", capfd.readouterr().err)]
    assert len(rounds) >= 3 and sum(k for _, k in rounds) >= 0.8 * sum(g + 1 for g, _ in rounds) and max(k for _, k in rounds) >= 4, rounds
    good = files["l6"]
    for kind in ("flip", "cut", "crc", "size"):
        blob = bytearray(good)
        if kind == "flip":
            blob[len(blob) * 3 // 4] ^= 0x10
        elif kind == "cut":
            blob = blob[: len(blob) * 2 // 3]
        elif kind == "crc":
            blob[-6] ^= 1
        else:
            blob[-2] ^= 1
        f = tmp_path / "bad.fastq.gz"
        f.write_bytes(bytes(blob))
        with pytest.raises((_lib.TbkError, ValueError, IOError, OSError)):
            read_all(f)


def test_guessing_inflate_side_by_side(built, tmp_path, monkeypatch):
    """Several gzip streams read at the same time (find-unique-kmers reads a library's files side by side):
    each reader gets its share of the host threads, the records are those of a reader alone."""
    import threading

    rng = np.random.default_rng(12)
    texts = []
    for f in range(3):
        n, length = 300 + 50 * f, 6000
        bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (n, length))
        texts.append(b"".join(b"@f%d_%d\n" % (f, i) + bases[i].tobytes() + b"\n+\n" + b"I" * length + b"\n" for i in range(n)))
        (tmp_path / f"lib{f}.fastq.gz").write_bytes(gzip.compress(texts[-1], 6))
    monkeypatch.setenv("TBK_PINFLATE_MIN", "0")
    monkeypatch.setenv("TBK_PINFLATE_SPAN", "100000")
    got = [None] * 3

    def read(f):
        got[f] = _native_records(str(tmp_path / f"lib{f}.fastq.gz"), max_bases=1 << 20)

    for rounds in range(2):
        threads = [threading.Thread(target=read, args=(f,)) for f in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for f in range(3):
            lines = texts[f].split(b"\n")
            assert got[f] == [[lines[4 * i][1:].decode(), lines[4 * i + 1].decode(), lines[4 * i + 3].decode()] for i in range(len(lines) // 4)]


def test_crc32_by_carryless_multiplication_equals_zlib(built):
    """tbk_crc.cpp: every length 0..300 and a few long ones, at three alignments, from two starting
    values; TBK_CRC=zlib (read once per process) is the fallback the library takes by itself when the CPU
    lacks PCLMULQDQ or its own known-answer check fails."""
    import ctypes as C
    import zlib

    from trio_binning_amd._lib import lib

    f = lib.tbk_crc32_c
    f.restype, f.argtypes = C.c_uint32, [C.c_uint32, C.c_char_p, C.c_size_t]
    data = np.random.default_rng(0).integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    for n in list(range(0, 300)) + [1000, 4095, 4096, 4097, 65537, (1 << 20) - 7]:
        for off in (0, 1, 7):
            piece = data[off:off + n]
            for start in (0, 0xDEADBEEF):
                assert f(start, piece, len(piece)) == zlib.crc32(piece, start), (n, off, start)
