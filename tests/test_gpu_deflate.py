"""The GPU gzip encoder of the bin writer (csrc/tbk_gdeflate.hip) against zlib: every member it writes must inflate to the
piece of text it was given - the reference's bins are read back through gzip (seq.py:86-92) and their DECOMPRESSED bytes are
the contract (seq.py:27-42,132-134; tests/test_classify_by_kmers.py:19-36), whichever encoder wrote them."""
import gzip
import hashlib
import os
import zlib
from unittest.mock import patch

import numpy as np
import pytest

from conftest import DATA, load_golden

pytestmark = pytest.mark.gpu


def fastq(rng, n_reads, length, qual):
    recs = []
    for i in range(n_reads):
        L = int(length if np.isscalar(length) else rng.integers(length[0], length[1]))
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].tobytes()
        if qual == "const":
            q = b"I" * L
        elif qual == "hifi":
            qv = np.clip(rng.normal(60, 15, L), 2, 93).astype(np.uint8)
            qv[rng.random(L) < 0.6] = 93
            q = (qv + 33).tobytes()
        else:
            q = (rng.integers(0, 4, L).astype(np.uint8) * 10 + 35).tobytes()   # binned
        recs.append(b"@read%d some comment\n" % i + seq + b"\n+\n" + q + b"\n")
    return b"".join(recs)


def check_members(pieces, members):
    assert len(members) == len(pieces)
    for i, (p, m) in enumerate(zip(pieces, members)):
        assert m[:4] == b"\x1f\x8b\x08\x00", i
        assert gzip.decompress(m) == p, (i, len(p), len(m))
        # one member, its trailer the piece's CRC-32 and length
        assert int.from_bytes(m[-8:-4], "little") == zlib.crc32(p) and int.from_bytes(m[-4:], "little") == len(p) & 0xFFFFFFFF
    # members back to back are one gzip file (what a bin is)
    assert gzip.decompress(b"".join(members)) == b"".join(pieces)


def test_members_inflate_to_their_text(gpu):
    from trio_binning_amd import seq

    rng = np.random.default_rng(11)
    pieces = [
        fastq(rng, 40, 15000, "hifi"),            # long lines: a block per line, bases and qualities coded apart
        fastq(rng, 40, 15000, "const"),           # runs: matches at distance 1
        fastq(rng, 3000, (50, 300), "binned"),    # short lines: 32 KiB blocks
        fastq(rng, 3, 200_000, "hifi"),           # lines longer than a block
        b"", b"A", b"\n", b"AB", b"A" * 100_000, bytes(range(256)) * 300,
        rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(),        # noise: stored blocks
        bytes(rng.integers(0, 2, 70_000, dtype=np.uint8) * 255),         # two symbols
        (b">r\n" + b"ACGT" * 20 + b"\n") * 2000,                          # FASTA, very regular
        b"".join(bytes([65 + (i % 7)]) * (i % 300 + 1) for i in range(2000)),   # runs of every length up to 300
    ]
    # a geometric frequency profile deep enough that a Huffman tree passes 15 levels (Fibonacci-like counts): the length limit
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    pieces.append(b"".join(bytes([33 + i]) * c for i, c in enumerate(fib))[:30000])
    members = seq.gzip_members_device(pieces)
    check_members(pieces, members)
    # the encoder shrinks what can be shrunk: 2 bits a base + the qualities' entropy, a few bits per 128-byte stretch of a run
    assert len(members[0]) < 0.50 * len(pieces[0]) and len(members[1]) < 0.16 * len(pieces[1])
    assert len(members[10]) < len(pieces[10]) + 5 * (len(pieces[10]) // 8192 + 2) + 64   # noise does not grow beyond the stored blocks' headers


@pytest.mark.parametrize("seed", range(int(os.environ.get("TBK_DEFLATE_FUZZ_SEEDS", "6"))))   # (a soak run: TBK_DEFLATE_FUZZ_SEEDS=400)
def test_fuzz_against_zlib(gpu, seed):
    from trio_binning_amd import seq

    rng = np.random.default_rng(1000 + seed)
    pieces = []
    for _ in range(60):
        kind = int(rng.integers(0, 6))
        n = int(rng.integers(1, 300_000)) if rng.random() < 0.8 else int(rng.integers(1, 40))
        if kind == 0:
            p = fastq(rng, max(1, n // 3000), (10, 6000), ["hifi", "const", "binned"][int(rng.integers(0, 3))])
        elif kind == 1:
            alphabet = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
            p = alphabet[rng.integers(0, alphabet.size, n)].tobytes()
        elif kind == 2:
            runs = rng.integers(1, 600, max(1, n // 100))
            vals = rng.integers(0, 256, runs.size, dtype=np.uint8)
            p = np.repeat(vals, runs).tobytes()
        elif kind == 3:
            w = 2.0 ** -np.arange(int(rng.integers(2, 60)))
            p = rng.choice(np.arange(w.size, dtype=np.uint8) + 40, size=n, p=w / w.sum()).tobytes()
        elif kind == 4:
            line = int(rng.integers(1, 5000))
            body = rng.integers(65, 91, n, dtype=np.uint8)
            body[line::line + 1] = 10
            p = body.tobytes()
        else:
            p = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        pieces.append(p)
    check_members(pieces, seq.gzip_members_device(pieces))


def test_a_gigabyte_of_fastq(gpu):
    """1 GB through the encoder in jobs of 128 members of 1 MiB, as the bin writer cuts them: sha256 of the inflated stream."""
    from trio_binning_amd import seq

    rng = np.random.default_rng(5)
    base = fastq(rng, 4600, 15000, "hifi")   # ~138 MB
    want, got = hashlib.sha256(), hashlib.sha256()
    total = 0
    for rep in range(8):
        text = base[rep * 1000:] + base[:rep * 1000]
        pieces = [text[i:i + (1 << 20)] for i in range(0, len(text), 1 << 20)]
        members = seq.gzip_members_device(pieces)
        d = zlib.decompressobj(31)
        for m in members:
            out = zlib.decompressobj(31).decompress(m)
            got.update(out)
        want.update(text)
        total += len(text)
    assert total > 1e9 and got.hexdigest() == want.hexdigest()


@pytest.mark.parametrize("encoder", ["gpu", "cpu"])
def test_cli_bins_through_either_encoder(gpu, capsys, tmp_path, monkeypatch, encoder):
    """classify-by-kmers with gzip'ed bins (the reference's default): the decompressed bins and the TSV equal the reference's
    recorded output whether the device or the host coded the members; TBK_STATS says which it was."""
    import trio_binning_amd.classify_by_kmers as cbk

    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == 21)
    fa, fb = tmp_path / "la.txt", tmp_path / "lb.txt"
    fa.write_text("".join(x + "\n" for x in v["list_a"]))
    fb.write_text("".join(x + "\n" for x in v["list_b"]))
    fq = tmp_path / "reads21.fa"
    with open(fq, "w") as fh:
        for i, s in enumerate(v["reads"]):
            fh.write(f">r{i} some comment\n{s}\n")
    monkeypatch.setenv("TBK_GZIP_ENCODER", encoder)
    monkeypatch.setenv("TBK_STATS", "1")
    monkeypatch.setattr(cbk, "_BATCH_BASES", 300)
    monkeypatch.setattr(cbk, "_BATCH_READS", 5)     # (dozens of flushes: the three-deep job ring turns over many times)
    od = tmp_path / "out"
    od.mkdir()
    with patch("sys.argv", ["classify-by-kmers", str(fq), str(fa), str(fb), "--haplotype-a-out-prefix", str(od / "hapA"),
                            "--haplotype-b-out-prefix", str(od / "hapB"), "--unclassified-out-prefix", str(od / "unclassified")]):
        cbk.main()
    out, err = capsys.readouterr()
    assert out == v["cli_stdout"]
    for fn, digest in v["cli_bins"].items():
        assert hashlib.sha256(gzip.open(od / fn, "rb").read()).hexdigest() == digest, fn
    assert ('"gzip_encoder": "%s"' % ("device" if encoder == "gpu" else "host")) in err, err[-600:]


def test_bin_writer_on_the_device_equals_the_python_writer(gpu, tmp_path):
    """seq.BinWriter(..., device=0): the three bins' gzip members coded on the GPU, many small batches (the job ring turns over, the
    bins' text changes buffers at every flush, members of 1 MiB are cut across batches), FASTQ and FASTA records mixed - the
    decompressed bins equal what the Python mirror of seq.py:27-42,98-136 writes; an empty bin is a valid empty gzip file."""
    import random

    from trio_binning_amd import seq

    rng = random.Random(19)
    recs = []
    for i in range(6000):
        n = rng.randrange(0, 2500)
        s = "".join(rng.choice("ACGT") for _ in range(n))
        kind = rng.random()
        q = None if kind < 0.3 else ("" if kind < 0.35 else "".join(rng.choice("!#5I~") for _ in range(n)))
        recs.append(seq.Read(f"read{i}/x", s, q))
    src = tmp_path / "in.fq"
    with open(src, "w") as fh:
        for r in recs:
            r.print(file=fh)
    for trial, letters in enumerate(("ABU", "AB")):   # (second trial: the unclassified bin stays empty)
        bins_all = "".join(rng.choice(letters) for _ in recs)
        outs = seq.open_outfiles(str(tmp_path / f"pa{trial}"), str(tmp_path / f"pb{trial}"), str(tmp_path / f"pu{trial}"), ".fq", True)
        parsed = list(seq.open_fastx_read(str(src)))
        for r, b in zip(parsed, bins_all):
            r.print(file=outs["ABU".index(b)])
        for fh in outs:
            fh.close()
        w = seq.BinWriter(str(tmp_path / f"na{trial}"), str(tmp_path / f"nb{trial}"), str(tmp_path / f"nu{trial}"), ".fq", True, device=0)
        assert w.gpu_encoder
        got_n = 0
        with seq.BatchReader(str(src)) as r:
            b = seq.Batch()
            while r.next_batch(b, 200_000, 0):
                w.write(b, bins_all[got_n:got_n + b.n_reads].encode())
                got_n += b.n_reads
        w.close()
        assert got_n == len(parsed)
        for py_name, nat_name in zip(seq.output_names(str(tmp_path / f"pa{trial}"), str(tmp_path / f"pb{trial}"), str(tmp_path / f"pu{trial}"), ".fq", True), w.names):
            assert gzip.open(nat_name, "rb").read() == gzip.open(py_name, "rb").read(), nat_name
