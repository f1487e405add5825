"""The GPU inflater of the reader's bgzf path (csrc/tbk_gdeflate.hip, second half) against zlib: the text of a bgzf file - what the
reference gets through gzip.open (seq.py:86-92) - must come out byte for byte, whatever deflate blocks its members hold (dynamic and
fixed Huffman codes, stored blocks, matches that overlap themselves, codes longer than the look-up tables' index), and a member that
is damaged must be refused, not returned."""
import gzip
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf(text: bytes, level=6, block=60000, strategy=zlib.Z_DEFAULT_STRATEGY, eof=True) -> bytes:
    out = bytearray()
    for i in range(0, len(text), block):
        blk = text[i:i + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        body = c.compress(blk) + c.flush()
        assert 18 + len(body) + 8 <= 65536
        out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk))
    return bytes(out) + (EOF_BLOCK if eof else b"")


def fastq(rng, n_reads, L, qual="hifi"):
    recs = []
    for i in range(n_reads):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].tobytes()
        if qual == "const":
            q = b"I" * L
        else:
            qv = np.clip(rng.normal(60, 15, L), 2, 93).astype(np.uint8)
            qv[rng.random(L) < 0.6] = 93
            q = (qv + 33).tobytes()
        recs.append(b"@read%d c\n" % i + seq + b"\n+\n" + q + b"\n")
    return b"".join(recs)


def test_bgzf_text_equals_zlibs(gpu):
    from trio_binning_amd import seq

    rng = np.random.default_rng(3)
    texts = {
        "hifi": fastq(rng, 60, 15000),
        "const": fastq(rng, 60, 15000, "const"),                        # runs: matches at distance 1 that overlap themselves
        "short reads": fastq(rng, 5000, 150),
        "noise": rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),   # incompressible: zlib stores
        "skewed": rng.choice(np.arange(200, dtype=np.uint8), size=400_000, p=(lambda w: w / w.sum())(1.5 ** -np.arange(200))).tobytes(),   # codes up to 15 bits: past the tables' index
        "periodic": (b"ACGTTGCA" * 7 + b"\n") * 8000,                     # long matches, small distances
        "one byte": b"x",
        "empty": b"",
    }
    for name, text in texts.items():
        for level, strategy, block in ((6, zlib.Z_DEFAULT_STRATEGY, 60000), (1, zlib.Z_DEFAULT_STRATEGY, 60000), (9, zlib.Z_DEFAULT_STRATEGY, 65280), (6, zlib.Z_FIXED, 30000),
                                       (0, zlib.Z_DEFAULT_STRATEGY, 50000), (6, zlib.Z_HUFFMAN_ONLY, 60000), (6, zlib.Z_RLE, 4000)):
            data = bgzf(text, level, block, strategy)
            assert gzip.decompress(data) == text
            assert seq.bgzf_inflate_device(data) == text, (name, level, strategy, block)
    # members separated by zero padding, no end-of-file block, an end-of-file block in the middle
    a, b = bgzf(texts["hifi"][:200_000], eof=False), bgzf(texts["const"][:100_000])
    assert seq.bgzf_inflate_device(a + b"\0" * 7 + EOF_BLOCK + b) == texts["hifi"][:200_000] + texts["const"][:100_000]
    assert seq.bgzf_inflate_device(EOF_BLOCK) == b""


def test_damaged_members_are_refused(gpu):
    from trio_binning_amd import seq
    from trio_binning_amd._lib import TbkError

    rng = np.random.default_rng(4)
    text = fastq(rng, 20, 15000)
    data = bytearray(bgzf(text))
    for where in (40, 3000, len(data) // 2, len(data) - 60):   # a flipped bit in a member's deflate stream: it fails to decode or fails its CRC-32
        bad = bytearray(data)
        bad[where] ^= 0x10
        with pytest.raises((TbkError, ValueError, IOError)):
            seq.bgzf_inflate_device(bytes(bad))
    wrong_crc = bytearray(data)
    wrong_crc[data.index(b"\x1f\x8b", 100) - 8] ^= 1           # the first member's CRC-32
    with pytest.raises((TbkError, ValueError, IOError)):
        seq.bgzf_inflate_device(bytes(wrong_crc))
    with pytest.raises((TbkError, ValueError, IOError)):
        seq.bgzf_inflate_device(bytes(data[: len(data) // 2]))   # cut inside a member
    with pytest.raises((TbkError, ValueError, IOError)):
        seq.bgzf_inflate_device(gzip.compress(text))             # an ordinary gzip member is not bgzf


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_against_zlib(gpu, seed):
    from trio_binning_amd import seq

    rng = np.random.default_rng(500 + seed)
    parts, want = [], []
    for _ in range(40):
        kind = int(rng.integers(0, 5))
        n = int(rng.integers(1, 400_000))
        if kind == 0:
            t = fastq(rng, max(1, n // 30000), 15000, ["hifi", "const"][int(rng.integers(0, 2))])
        elif kind == 1:
            alphabet = rng.integers(0, 256, int(rng.integers(1, 60)), dtype=np.uint8)
            t = alphabet[rng.integers(0, alphabet.size, n)].tobytes()
        elif kind == 2:
            runs = rng.integers(1, 700, max(1, n // 200))
            t = np.repeat(rng.integers(0, 256, runs.size, dtype=np.uint8), runs).tobytes()
        elif kind == 3:
            unit = rng.integers(65, 91, int(rng.integers(1, 300)), dtype=np.uint8).tobytes()
            t = (unit * (n // len(unit) + 1))[:n]
        else:
            t = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        parts.append(bgzf(t, int(rng.integers(0, 10)), int(rng.integers(1000, 65281)),
                          [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED][int(rng.integers(0, 5))], eof=bool(rng.integers(0, 2))))
        want.append(t)
    assert seq.bgzf_inflate_device(b"".join(parts)) == b"".join(want)


def test_reader_and_cli_from_bgzf_on_the_device(gpu, capfd, tmp_path, monkeypatch):
    """The reader's BGZF path on the GPU (tbk_fastx_set_device): the records of a .fastq.gz written as BGZF come out as from the
    plain file, over many small windows (TBK_BGZF_GPU_WINDOW: the three-slot ring turns over, windows end between blocks, the
    end-of-file block is a window of its own), with an ordinary gzip member behind the blocks (the host path takes over there);
    and classify-by-kmers on it writes the reference's recorded TSV and bins."""
    import hashlib
    from unittest.mock import patch

    import trio_binning_amd.classify_by_kmers as cbk
    from conftest import load_golden
    from trio_binning_amd import seq

    rng = np.random.default_rng(8)
    text = fastq(rng, 300, 4000)
    plain = tmp_path / "r.fastq"
    plain.write_bytes(text)

    def records(path, **kw):
        out = []
        with seq.BatchReader(str(path), **kw) as r:
            on_device = r.inflates_on_device
            b = seq.Batch()
            while r.next_batch(b, 200_000, 0):
                out += [(x.name, x.seq, x.qual) for x in b.reads()]
        return out, on_device

    want, _ = records(plain)
    assert len(want) == 300
    monkeypatch.setenv("TBK_BGZF_GPU_WINDOW", "70000")
    for name, data in (("blocks", bgzf(text, 6, 20000)), ("no_eof", bgzf(text, 1, 65280, eof=False)),
                       ("member_behind", bgzf(text[:1_000_000], 6, 30000, eof=False) + gzip.compress(text[1_000_000:]))):
        path = tmp_path / f"{name}.fastq.gz"
        path.write_bytes(data)
        # the windows parsed where the device wrote them (what the parser had left copied in front), copied into the reader's own
        # buffer (no room in front), and both by turns (room for some of the leftovers only)
        for room in (None, "0", "6000"):
            if room is None:
                monkeypatch.delenv("TBK_BGZF_GPU_ROOM", raising=False)
            else:
                monkeypatch.setenv("TBK_BGZF_GPU_ROOM", room)
            got, on_device = records(path, device=0)
            assert on_device and got == want, (name, room)
        monkeypatch.delenv("TBK_BGZF_GPU_ROOM", raising=False)
        # borrowing (what the native loop asks for): a batch's records stay in the inflater's window, which stays out of the ring until the
        # last batch that refers to it is refilled or destroyed - kept here beyond the reader
        with seq.BatchReader(str(path), packing=True, borrowing=True, device=0) as r:
            kept, borrowed = [], 0
            while True:
                b = seq.Batch()
                if not r.next_batch(b, 200_000, 0):
                    break
                borrowed += b.borrowed
                kept.append(b)   # every batch of the file: more than the ring has windows - those that find it short of free ones copy their records
        assert (0 if name == "member_behind" else 1) <= borrowed < len(kept), (name, borrowed, len(kept))   # (the first batch of member_behind reads on into the ordinary member: copied)
        assert [(x.name, x.seq, x.qual) for b in kept for x in b.reads()] == want, name   # (the reader is closed)
        del kept
        # a ring of fewer windows (as if the pinned memory for more were not there): two or three turn over; with one the host's threads inflate
        for slots in ("1", "2", "3"):
            monkeypatch.setenv("TBK_BGZF_GPU_SLOTS", slots)
            got, _ = records(path, device=0)
            assert got == want, (name, slots)
            with seq.BatchReader(str(path), packing=True, borrowing=True, device=0) as r:
                got, b = [], seq.Batch()
                while r.next_batch(b, 200_000, 0):
                    got += [(x.name, x.seq, x.qual) for x in b.reads()]
            assert got == want, (name, slots)
        monkeypatch.delenv("TBK_BGZF_GPU_SLOTS", raising=False)
        got_cpu, on_device = records(path)
        assert not on_device and got_cpu == want, name
    # a damaged block: the run fails, it does not go on with wrong text
    bad = bytearray(bgzf(text, 6, 20000))
    bad[len(bad) // 2] ^= 0x20
    (tmp_path / "bad.fastq.gz").write_bytes(bytes(bad))
    with pytest.raises(Exception):
        records(tmp_path / "bad.fastq.gz", device=0)

    # the command line: the reference's recorded output from a BGZF copy of the golden reads
    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == 21)
    fa, fb = tmp_path / "la.txt", tmp_path / "lb.txt"
    fa.write_text("".join(x + "\n" for x in v["list_a"]))
    fb.write_text("".join(x + "\n" for x in v["list_b"]))
    fq = tmp_path / "reads21.fa.gz"
    fq.write_bytes(bgzf("".join(f">r{i} some comment\n{s}\n" for i, s in enumerate(v["reads"])).encode(), 6, 3000))
    monkeypatch.setenv("TBK_BGZF_GPU_WINDOW", "8000")
    monkeypatch.setenv("TBK_WRITE_TIMING", "1")
    monkeypatch.setattr(cbk, "_BATCH_BASES", 3000)
    monkeypatch.setattr(cbk, "_BATCH_READS", 20)
    for devices in ("0", "0,0"):   # (two rings on the device beside the inflater and the bins' encoder)
        monkeypatch.setenv("TBK_DEVICES", devices)
        od = tmp_path / ("out" + devices.replace(",", "_"))
        od.mkdir()
        with patch("sys.argv", ["classify-by-kmers", str(fq), str(fa), str(fb), "--haplotype-a-out-prefix", str(od / "hapA"),
                                "--haplotype-b-out-prefix", str(od / "hapB"), "--unclassified-out-prefix", str(od / "unclassified")]):
            cbk.main()
        out, err = capfd.readouterr()
        assert out == v["cli_stdout"], devices
        for fn, digest in v["cli_bins"].items():
            assert hashlib.sha256(gzip.open(od / fn, "rb").read()).hexdigest() == digest, (fn, devices)
        assert "tbk-gpu-bgzf" in err and " 0 windows" not in err, err[-500:]
