"""Lists from files on the GPU box: the GPU parser (regular lists), the general host parser (everything
else) and the binary key cache give the same keys - those `tbk_list_parse_file` (host only, compared with the
real reference's tables in tests/test_oracle_golden.py) produces (c/kmers.c:124-229)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_list(path, rng, n, k, last_newline=True, odd=True):
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    rows = alphabet[rng.integers(0, 4, (n, k))]
    if odd:  # bytes outside ACGT pack as 0 (c/kmers.c:54-68): lower case, N, a CR
        for i in rng.integers(0, n, 50):
            rows[i, int(rng.integers(0, k))] = rng.choice(np.frombuffer(b"acgtNn-\r", dtype=np.uint8))
    text = b"\n".join(r.tobytes() for r in rows) + (b"\n" if last_newline else b"")
    path.write_bytes(text)
    return text


@pytest.mark.parametrize("k,last_newline", [(21, True), (21, False), (5, True), (32, True), (31, False), (1, True)])
def test_gpu_parser_equals_the_host_parser(gpu, tmp_path, monkeypatch, k, last_newline):
    from trio_binning_amd import kmers

    monkeypatch.delenv("TBK_LIST_CACHE", raising=False)
    rng = np.random.default_rng(k)
    p = tmp_path / "list.txt"
    _write_list(p, rng, 300_000 if k > 1 else 3, k, last_newline)
    want, k_host = kmers.parse_kmer_list(str(p))
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "gpu-parser" and hs.k == k_host == k and hs.num_kmers == want.size
        assert np.array_equal(hs.keys(), want)
    monkeypatch.setenv("TBK_LIST_GPU_PARSE", "0")
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "keys" and np.array_equal(hs.keys(), want)


def test_irregular_lists_go_to_the_general_parser(gpu, tmp_path):
    """Anything but k bytes + newline per line - a short line, a long line, a blank last line - is the general
    parser's (the reference's getline rules); the GPU parser notices and steps aside."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(3)
    base = _write_list(tmp_path / "r.txt", rng, 5000, 21, True, odd=False).split(b"\n")[:-1]
    cases = {
        "short_line_same_size": base[:100] + [base[100][:20], base[101] + b"A"] + base[102:],   # total size still n * 22
        "blank_last_line": base + [b""],
        "long_line": base[:7] + [base[7] + b"ACGTACGTACGTACGTACGTAC"] + base[8:],               # adds a whole stride
        "crlf": [x + b"\r" for x in base],
    }
    for name, lines in cases.items():
        p = tmp_path / (name + ".txt")
        p.write_bytes(b"\n".join(lines) + b"\n")
        want, k_host = kmers.parse_kmer_list(str(p))
        with kmers.HashSet.from_file(str(p), 0) as hs:
            assert hs.k == k_host and hs.num_kmers == want.size, name
            assert np.array_equal(hs.keys(), want), name
            assert hs.origin == ("gpu-parser" if name == "crlf" else "keys"), (name, hs.origin)   # CRLF lines are regular lines of k + 1 bytes


def test_key_cache_round_trip(gpu, tmp_path, monkeypatch):
    from trio_binning_amd import kmers

    rng = np.random.default_rng(9)
    p = tmp_path / "list.txt"
    _write_list(p, rng, 200_000, 21)
    want, _ = kmers.parse_kmer_list(str(p))
    cache = str(p) + ".tbk"
    monkeypatch.delenv("TBK_LIST_CACHE", raising=False)
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "gpu-parser" and not os.path.exists(cache)          # no cache is written unasked
    monkeypatch.setenv("TBK_LIST_CACHE", "1")
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "gpu-parser" and os.path.getsize(cache) == 56 + 8 * want.size
    monkeypatch.delenv("TBK_LIST_CACHE")
    with kmers.HashSet.from_file(str(p), 0) as hs:                                # a valid cache is used without being asked for
        assert hs.origin == "cache" and hs.k == 21 and np.array_equal(hs.keys(), want)
    monkeypatch.setenv("TBK_LIST_CACHE", "0")
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "gpu-parser"
    monkeypatch.delenv("TBK_LIST_CACHE")
    # damaged payload: ignored, the text is parsed
    raw = bytearray(open(cache, "rb").read())
    raw[56 + 8 * 1234] ^= 0x40
    st = os.stat(p)
    open(cache, "wb").write(bytes(raw))
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "gpu-parser" and np.array_equal(hs.keys(), want)
    raw[56 + 8 * 1234] ^= 0x40
    open(cache, "wb").write(bytes(raw))
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "cache"
    # another list of the same k and line count (= the same size to the byte) under the old name, with the old
    # modification time (cp -p, rsync -t, tar): the text's fingerprint no longer matches, the cache is not used
    other = tmp_path / "other.txt"
    _write_list(other, np.random.default_rng(10), 200_000, 21)
    assert os.path.getsize(other) == os.path.getsize(p)
    keep = open(p, "rb").read()
    open(p, "wb").write(open(other, "rb").read())
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns))
    want_other, _ = kmers.parse_kmer_list(str(p))
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "gpu-parser" and np.array_equal(hs.keys(), want_other) and not np.array_equal(want_other, want)
    open(p, "wb").write(keep)
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns))
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "cache" and np.array_equal(hs.keys(), want)
    # the list changes (size and time): the cache no longer belongs to it
    with open(p, "ab") as fh:
        fh.write(b"A" * 21 + b"\n")
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
    want2, _ = kmers.parse_kmer_list(str(p))
    with kmers.HashSet.from_file(str(p), 0) as hs:
        assert hs.origin == "gpu-parser" and hs.num_kmers == want.size + 1 and np.array_equal(hs.keys(), want2)
