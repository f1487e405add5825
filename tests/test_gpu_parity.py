"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle on the
same inputs — bit-exact (integer counts).  Marked gpu: they need a real MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import DATA, load_golden

pytestmark = pytest.mark.gpu

COMP = str.maketrans("ACGT", "TGCA")


@pytest.fixture(autouse=True, params=["packed-h2d", "ascii-h2d"])
def transfer(request, monkeypatch):
    """Every test of this file runs twice: host batches cross PCIe in the packed transfer format
    (the default: packed on the host, the kernel starts from code words and masks) and as ASCII
    (TBK_PACKED_H2D=0: the kernel packs).  Same oracle, same expected counts."""
    monkeypatch.setenv("TBK_PACKED_H2D", "1" if request.param == "packed-h2d" else "0")
    return request.param


def _rc(s):
    return s.translate(COMP)[::-1]


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def _pack(reads):
    from trio_binning_amd import kmers

    return kmers.pack_reads(reads)


def _rand_reads(rng, n_reads, max_len, plants, k, p_plant=0.7):
    reads = []
    for _ in range(n_reads):
        n = int(rng.integers(0, max_len + 1))
        s = list("".join("ACGT"[c] for c in rng.integers(0, 4, n)))
        if plants and n >= k and rng.random() < p_plant:
            for _ in range(int(rng.integers(1, 8))):
                km = plants[int(rng.integers(0, len(plants)))]
                if rng.random() < 0.5:
                    km = _rc(km)
                p = int(rng.integers(0, n - k + 1))
                s[p:p + k] = km
        reads.append("".join(s))
    return reads


# ---- reference KATs through the product API -------------------------------------------------
def test_reference_kats(gpu, capfd):
    # reference tests/test_kmers.py:8-25
    from trio_binning_amd import kmers

    a = kmers.create_kmer_hash_set(os.path.join(DATA, "hapA.txt"))
    b = kmers.create_kmer_hash_set(os.path.join(DATA, "hapB.txt"))
    assert kmers.get_number_kmers_in_set(a) == 4
    assert kmers.get_number_kmers_in_set(b) == 3
    assert a.k == 21 and a.contents.num_kmers == 4
    read = "CTTATCATGTCTTTGTTTTCAAAGCTTCTTAGAGGTTTTTTTTTTTGGTGTTAATTGGCATAAATTATGGCT"
    assert kmers.count_kmers_in_read(read, a, b) == (2, 1)
    g = load_golden("kat.json")
    for case in g["count_kmers_in_read"]:
        assert list(kmers.count_kmers_in_read(case["read"], a, b)) == case["counts"]


@pytest.mark.parametrize("idx", range(6))
def test_golden_differential_vectors(gpu, tmp_path, idx):
    """Counts recorded from the real reference, k in {5,13,21,27,31,32}, with duplicate,
    in-both-lists and non-canonical list lines."""
    from trio_binning_amd import kmers

    v = load_golden("diff_vectors.json")[idx]
    a = kmers.HashSet.from_file(_write(tmp_path, "a.txt", "".join(x + "\n" for x in v["list_a"])))
    b = kmers.HashSet.from_file(_write(tmp_path, "b.txt", "".join(x + "\n" for x in v["list_b"])))
    assert [a.num_kmers, b.num_kmers] == v["num_kmers"] and a.k == v["k"]
    with kmers.Classifier(a, b) as cls:
        got = cls.classify_reads(v["reads"])
    assert got.tolist() == v["counts"]
    # the single-read entry point too (first 25 reads)
    for r, c in list(zip(v["reads"], v["counts"]))[:25]:
        assert list(kmers.count_kmers_in_read(r, a, b)) == c


def test_golden_edge_vectors(gpu, tmp_path):
    from trio_binning_amd import kmers

    for case in load_golden("edge_vectors.json"):
        a = kmers.HashSet.from_file(_write(tmp_path, "a.txt", case["text_a"]))
        b = kmers.HashSet.from_file(_write(tmp_path, "b.txt", case["text_b"]))
        assert [a.num_kmers, b.num_kmers] == case["num_kmers"], case["name"]
        assert a.k == case["k"], case["name"]
        with kmers.Classifier(a, b) as cls:
            got = cls.classify_reads(case["reads"])
        assert got.tolist() == case["counts"], case["name"]


@pytest.mark.parametrize("layout,k", [("auto", 5), ("keys", 21), ("entries", 21), ("short_keys", 21), ("full_keys", 31), ("wide_entries", 31), ("auto", 32)])
def test_one_read_per_call_equals_the_oracle(gpu, orc, tmp_path, layout, k):
    """count_kmers_in_read (c/kmers.c:270-299) called the way the reference's driver calls it - once per read,
    classify_by_kmers.py:99-102 - takes a path of its own (tbk_host.cpp count_small: the read is read by the single-read
    kernel from pinned host memory, the counters run on between calls): every read of the golden vector, reads shorter than
    k, of exactly k bases, with bytes outside ACGT, a read long enough to make the path's buffer grow and one beyond its
    limit (the batch path), in an order that mixes them - each count equal to the oracle's."""
    from trio_binning_amd import kmers

    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == k)
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in v["list_a"]))
    fb = _write(tmp_path, "b.txt", "".join(x + "\n" for x in v["list_b"]))
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    rng = np.random.default_rng(77 + k)
    keys = v["list_a"] + v["list_b"]
    long_read = "".join("ACGT"[c] for c in rng.integers(0, 4, 1_600_000))
    for pos in range(1000, 1_590_000, 50_021):
        long_read = long_read[:pos] + keys[pos % len(keys)] + long_read[pos + k:]
    huge = long_read * 6                       # 9.6 Mb: beyond the one-launch path's 8 Mi bases
    reads = list(v["reads"]) + ["", "A", keys[0][: k - 1], keys[0], _rc(keys[1]), keys[2] + "N" + keys[3], "acgt" * 20 + keys[4], long_read, keys[5] * 3, huge, v["reads"][0]]
    order = rng.permutation(len(reads))
    bases, offs = _pack([reads[i] for i in order])
    want = orc.count_batch(bases, offs, oa, ob)
    with kmers.Classifier(a, b, options=kmers.Options.layout(layout)) as cls:
        st = cls.stats()
        assert {"keys": not (st["entry_layout"] or st["short_keys"] or st["full_keys"]), "entries": st["entry_layout"] and not st["wide_entries"], "wide_entries": st["wide_entries"],
                "short_keys": st["short_keys"], "full_keys": st["full_keys"], "auto": True}[layout], st
        for j, i in enumerate(order):
            assert list(cls.count_read(reads[i])) == want[j].tolist(), (j, i, len(reads[i]))
        # interleaved with batches on the same classifier: the running counters are the one-read path's own
        assert np.array_equal(cls.classify_batch(bases, offs), want)
        assert list(cls.count_read(reads[0])) == want[list(order).index(0)].tolist()
    for j, i in enumerate(order[:60]):      # (the reference-named entry point: its cached classifier is built with the defaults)
        assert list(kmers.count_kmers_in_read(reads[i], a, b)) == want[j].tolist(), (j, i, len(reads[i]))
    assert want.sum() > 100
    # the same list again after the lists changed places: the cached classifier follows the arguments
    want_ba = orc.count_batch(bases, offs, ob, oa)
    for j, i in enumerate(order[:40]):
        assert list(kmers.count_kmers_in_read(reads[i], b, a)) == want_ba[j].tolist(), (j, i)


# ---- seeded random inputs against the oracle ----------------------------------------------------
@pytest.mark.parametrize("k", [1, 2, 3, 5, 15, 16, 17, 21, 27, 31, 32])
def test_random_vs_oracle(gpu, orc, tmp_path, k):
    from trio_binning_amd import kmers

    rng = np.random.default_rng(1000 + k)
    n_list = min(200, 4 ** k // 2 + 2)
    la = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(n_list)]
    lb = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(n_list)] + la[:2]
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    fb = _write(tmp_path, "b.txt", "\n".join(lb))
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    assert (a.num_kmers, b.num_kmers, a.k) == (oa.num_kmers, ob.num_kmers, oa.k)
    reads = _rand_reads(rng, 300, 3000, la + lb, k)
    reads += ["", "A", "ACGT" * 300, la[0], _rc(la[0]), la[0] * 3]
    bases, offs = _pack(reads)
    with kmers.Classifier(a, b) as cls:
        got = cls.classify_batch(bases, offs)
    want = orc.count_batch(bases, offs, oa, ob)
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:10]
    assert want.sum() > 0


@pytest.mark.parametrize("mod_sampling", [1, 0])
@pytest.mark.parametrize("w", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_every_bucket_selection_mode(gpu, orc, tmp_path, w, mod_sampling, monkeypatch):
    """Bucket selection (plain hash; spans of 1..8 m-mers sampled by mod-sampling or by the
    random-minimizer rule) is a layout choice: counts must not depend on it.  k = 21 / 22 /
    31 / 32 exercise both m-mer lengths and non-zero span offsets."""
    from trio_binning_amd import kmers

    monkeypatch.setenv("TBK_MINIMIZER_W", str(w))
    monkeypatch.setenv("TBK_MOD_SAMPLING", str(mod_sampling))
    rng = np.random.default_rng(77 + w)
    for k in (21, 22, 31, 32):
        la = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(300)]
        lb = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(300)]
        la += ["A" * k, "ACGT" * 8][:2] if k == 32 else ["A" * k]
        fa = _write(tmp_path, "a.txt", "".join(x[:k] + "\n" for x in la))
        fb = _write(tmp_path, "b.txt", "".join(x + "\n" for x in lb))
        oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
        a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
        reads = _rand_reads(rng, 200, 2500, la + lb, k, p_plant=0.9) + ["A" * 100, "T" * 100, "AC" * 60]
        # tie-heavy reads: short-period repeats and reverse-complement palindromes of list k-mers
        reads += [("ACGTTGCA" * 20)[:150], "ATAT" * 30 + la[0] + "ATAT" * 30, la[1] + _rc(la[1]) + lb[1] + _rc(lb[1]), "G" * 50 + lb[0] + "C" * 50]
        bases, offs = _pack(reads)
        with kmers.Classifier(a, b) as cls:
            st = cls.stats()
            assert st["minimizer_w"] <= w and (w == 0) == (st["minimizer_w"] == 0)
            if st["minimizer_w"]:
                span = st["minimizer_m"] + st["minimizer_w"] - 1
                assert span <= k and (k - span) % 2 == 0 and st["span_offset"] == (k - span) // 2
            t_expect = st["minimizer_m"] - st["minimizer_w"]
            if not mod_sampling or st["minimizer_w"] < 2 or t_expect < 8:
                assert st["sampling_t"] == 0
            else:
                assert st["sampling_t"] == t_expect
            got = cls.classify_batch(bases, offs)
        want = orc.count_batch(bases, offs, oa, ob)
        assert np.array_equal(got, want), (k, w, st, np.nonzero((got != want).any(axis=1))[0][:10])
        assert want.sum() > 100


@pytest.mark.parametrize("m", [17, 18, 20, 23, 26])
def test_long_mmer_bucket_selection(gpu, orc, tmp_path, m, monkeypatch):
    """Big tables select buckets by minimizers longer than 16 bases (64-bit m-mer path);
    force that path at test size.  Counts must not change."""
    from trio_binning_amd import kmers

    monkeypatch.setenv("TBK_MINIMIZER_M", str(m))
    monkeypatch.setenv("TBK_MOD_SAMPLING", str(m % 2))  # both sampling rules on the 64-bit m-mer path
    rng = np.random.default_rng(300 + m)
    for k, w in ((27, 6), (31, 6), (32, 5), (31, 3)):
        if m + 1 > k:
            continue
        monkeypatch.setenv("TBK_MINIMIZER_W", str(w))
        la = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(300)] + ["A" * k, "AC" * 16]
        lb = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(300)]
        fa = _write(tmp_path, "a.txt", "".join(x[:k] + "\n" for x in la))
        fb = _write(tmp_path, "b.txt", "".join(x + "\n" for x in lb))
        oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
        a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
        reads = _rand_reads(rng, 200, 2500, [x[:k] for x in la] + lb, k, p_plant=0.9) + ["A" * 100, "T" * 100, "AC" * 60]
        bases, offs = _pack(reads)
        with kmers.Classifier(a, b) as cls:
            st = cls.stats()
            if st["minimizer_w"]:
                assert st["minimizer_m"] == m
                span = m + st["minimizer_w"] - 1
                assert span <= k and (k - span) % 2 == 0
            got = cls.classify_batch(bases, offs)
        want = orc.count_batch(bases, offs, oa, ob)
        assert np.array_equal(got, want), (k, m, st, np.nonzero((got != want).any(axis=1))[0][:10])
        assert want.sum() > 100


@pytest.mark.parametrize("seed", range(int(os.environ.get("TBK_FUZZ_SEEDS", "24"))))  # more seeds for a soak run
def test_seeded_fuzz_of_layout_and_input_shapes(gpu, orc, tmp_path, seed, monkeypatch):
    """Random small configurations: any k, lists with duplicate / shared / reverse-complement /
    low-complexity lines, reads with N and lower case, ragged lengths (empty reads, reads shorter
    than k, reads ending inside and exactly on a 2048-window pass), every bucket-selection mode and
    table loads from roomy to crowded.  Counts must equal the oracle's in all of them."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([1, 2, 4, 7, 11, 14, 15, 16, 17, 19, 21, 22, 23, 25, 27, 29, 31, 32]))
    monkeypatch.setenv("TBK_MINIMIZER_W", str(int(rng.integers(0, 9))))
    monkeypatch.setenv("TBK_MOD_SAMPLING", str(int(rng.integers(0, 2))))
    monkeypatch.setenv("TBK_TABLE_LOAD", str(rng.choice([0.04, 0.2, 0.6, 0.9])))
    monkeypatch.setenv("TBK_GUESTS", str(seed % 3 and 1))     # a third of the seeds without guests in the other half
    monkeypatch.setenv("TBK_SLICE_BASES", str(int(rng.choice([2048, 5000, 1 << 30]))))  # an empty ring takes a batch in up to 8 slices
    monkeypatch.setenv("TBK_FRONT", str(seed // 3 % 2))       # half of them in the front layout (where mod-sampling is drawn): crowded fronts, walks from the home line
    n_a, n_b = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))

    def rand_kmer():
        mode = rng.random()
        if mode < 0.1:   # low complexity: long runs of one base
            return ("ACGT"[int(rng.integers(0, 4))] * k)[: int(rng.integers(0, k + 1))].ljust(k, "ACGT"[int(rng.integers(0, 4))])
        if mode < 0.2:   # short-period repeat
            unit = "".join("ACGT"[c] for c in rng.integers(0, 4, int(rng.integers(1, 4))))
            return (unit * k)[:k]
        return "".join("ACGT"[c] for c in rng.integers(0, 4, k))

    la = [rand_kmer() for _ in range(n_a)]
    lb = [rand_kmer() for _ in range(n_b)]
    lb += [la[int(i)] for i in rng.integers(0, n_a, min(20, n_a))]          # shared with hapA
    lb += [_rc(la[int(i)]) for i in rng.integers(0, n_a, min(10, n_a))]     # shared, other strand
    la += [la[int(i)] for i in rng.integers(0, n_a, 5)]                     # duplicates
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    fb = _write(tmp_path, "b.txt", "\n".join(lb))                          # no trailing newline
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    assert (a.num_kmers, b.num_kmers) == (oa.num_kmers, ob.num_kmers)

    reads = _rand_reads(rng, int(rng.integers(1, 120)), int(rng.choice([40, 300, 2500, 9000])), la + lb, k, p_plant=0.9)
    reads += ["", "A" * max(0, k - 1), la[0], _rc(lb[0]) * 2, "".join(la[:40]), "".join(_rc(x) for x in lb[:40])]
    body = "".join("ACGT"[c] for c in rng.integers(0, 4, 5000))
    reads += [body[: 2048 - sum(map(len, reads)) % 2048], body[:2047], body[:2048 + k - 1], body[:4096]]  # pass-boundary shapes
    noisy = list(body[:3000])
    for i in rng.integers(0, 3000, 40):
        noisy[int(i)] = "NnacgtR-"[int(rng.integers(0, 8))]
    reads.append("".join(noisy))
    order = rng.permutation(len(reads))
    reads = [reads[int(i)] for i in order]
    bases, offs = _pack(reads)
    want = orc.count_batch(bases, offs, oa, ob, strict=True)
    with kmers.Classifier(a, b) as cls:
        got = cls.classify_batch(bases, offs)
        again = cls.classify_batch(bases, offs)
    assert np.array_equal(got, want), (seed, k, cls_env(), np.nonzero((got != want).any(axis=1))[0][:10])
    assert np.array_equal(again, want)


def cls_env():
    return {v: os.environ.get(v) for v in ("TBK_MINIMIZER_W", "TBK_MOD_SAMPLING", "TBK_TABLE_LOAD")}


def test_ragged_batch_shapes(gpu, orc, tmp_path):
    """Empty batch, empty reads, many tiny reads, reads around the 1024-window pass size and
    the 16-base chunk size, one long read: per-read attribution at every boundary."""
    from trio_binning_amd import kmers

    k = 21
    rng = np.random.default_rng(5)
    la = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(50)]
    lb = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(50)]
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    fb = _write(tmp_path, "b.txt", "".join(x + "\n" for x in lb))
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)

    def dense(n):  # a read made of list k-mers back to back, both strands
        s = ""
        while len(s) < n:
            km = (la + lb)[int(rng.integers(0, 100))]
            s += km if rng.random() < 0.5 else _rc(km)
        return s[:n]

    shapes = {
        "empty_batch": [],
        "only_empty_reads": ["", "", ""],
        "tiny_reads": [dense(int(n)) for n in rng.integers(0, 60, 2000)],
        "around_pass": [dense(n) for n in (1003, 1004, 1023, 1024, 1025, 1044, 1045, 2047, 2048, 2049, 15, 16, 17, 31, 32, 33)],
        "one_long": [dense(300_000)],
        "mixed": [dense(int(n)) for n in rng.integers(0, 5000, 300)] + [""] * 5 + [dense(70_000)] + [dense(20), dense(21)],
    }
    with kmers.Classifier(a, b) as cls:
        for name, reads in shapes.items():
            bases, offs = _pack(reads)
            got = cls.classify_batch(bases, offs)
            want = orc.count_batch(bases, offs, oa, ob)
            assert got.shape == want.shape, name
            assert np.array_equal(got, want), (name, np.nonzero((got != want).any(axis=1))[0][:10])


def test_non_acgt_policy(gpu, orc, tmp_path):
    """Documented deviation (reference is undefined there): a window containing a byte
    outside ACGT scores no hit; the oracle implements the same policy."""
    from trio_binning_amd import kmers

    fa, fb = os.path.join(DATA, "hapA.txt"), os.path.join(DATA, "hapB.txt")
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    km_a, km_b = "ACCTCTAAGAAGCTTTGAAAA", "AACACCAAAAAAAAAAACCTC"
    rng = np.random.default_rng(9)
    reads = [km_a, "N" + km_a + "N", km_a[:10] + "N" + km_a[11:], km_a.lower(), km_a + "n" + km_b,
             km_a + "\n" + km_b, "*" * 50, km_b + "-" + km_b[::-1]]
    for _ in range(200):
        n = int(rng.integers(30, 500))
        s = list("".join("ACGTNacgt*-"[c] for c in rng.choice(11, n, p=[.22, .22, .22, .22, .04, .02, .02, .01, .01, .01, .01])))
        p = int(rng.integers(0, n - 21))
        s[p:p + 21] = km_a if rng.random() < 0.5 else _rc(km_b)
        reads.append("".join(s))
    bases, offs = _pack(reads)
    with kmers.Classifier(a, b) as cls:
        got = cls.classify_batch(bases, offs)
    want = orc.count_batch(bases, offs, oa, ob, strict=True)
    assert np.array_equal(got, want)
    assert got[0].tolist() == [1, 0] and got[2].tolist() == [0, 0] and got[3].tolist() == [0, 0]


def test_list_parser_rules(gpu, orc, tmp_path):
    """peek_at_file / getline rules (c/kmers.c:124-146): k from the first line, duplicates
    counted, last line without newline counted, over-long lines use their first k bytes,
    bytes outside ACGT pack as 0 ('A'), CRLF lists get k+1."""
    from trio_binning_amd import kmers

    cases = {
        "dups": "ACGTACGTAC\nACGTACGTAC\nTTTTTTTTTT\nGGGGGGGGGG\n",
        "no_final_newline": "ACGTACGTAC\nCCCCCCCCCC\nGGGTGGGTGG\nAAAAAAAAAC",
        "long_lines": "ACGTACGTAC\nACGTACGTACGGGG\nTTTTTTTTTTA\nCCCCCCCCCC\n",
        "non_acgt_in_list": "ACGTACGTAC\nACGNACGTAC\nacgtacgtac\nCCCCCCCCCC\n",
        "crlf": "ACGTACGTAC\r\nCCCCCCCCCC\r\nGGGGGGGGGG\r\nTTTTTTTTTA\r\n",
        "k_equals_line_with_newline": "ACGT\nACG\nCCCC\nGGGG\n",
        # lines shorter than k pack the reference's getline buffer (the line, a NUL, earlier lines' tail)
        "blank_last_line": "ACGTACGTAC\nCCCCCCCCCC\n\n",
        "short_lines": "ACGTACGTAC\nACG\nAAAAAAAAAA\nGG\n\nTTTTTTTTTA\nC",
    }
    rng = np.random.default_rng(3)
    reads = ["".join("ACGT"[c] for c in rng.integers(0, 4, 200)) for _ in range(50)]
    reads += ["ACGTACGTAC", "ACGAACGTAC", "AAAAAAAAAA", "ACGTACGTACA", "GGGGGGGGGGA", "ACGA", "CCCCA", "TTTTTTTTTAA"]
    bases, offs = _pack(reads)
    for name, text in cases.items():
        f = _write(tmp_path, name + ".txt", text)
        oa, a = orc.table_from_file(f), kmers.HashSet.from_file(f)
        assert (a.k, a.num_kmers) == (oa.k, oa.num_kmers), name
        # a hapB list of the same k (lists of different k cannot be paired)
        other = _write(tmp_path, "other.txt", "".join(("GATTACAGATTACA"[:oa.k - 1] + c).ljust(oa.k, "C") + "\n" for c in "ACGT"))
        ob, b = orc.table_from_file(other), kmers.HashSet.from_file(other)
        assert ob.k == oa.k
        with kmers.Classifier(a, b) as cls:
            got = cls.classify_batch(bases, offs)
        assert np.array_equal(got, orc.count_batch(bases, offs, oa, ob)), name
    for bad in ("", "A" * 33 + "\n"):
        with pytest.raises(ValueError):
            kmers.HashSet.from_file(_write(tmp_path, "bad.txt", bad))
    with pytest.raises(IOError):
        kmers.HashSet.from_file(str(tmp_path / "missing.txt"))


def test_mismatched_k_is_rejected(gpu, tmp_path):
    """The reference takes the window length from hapA's k but packs hapB lookups with
    hapB's k (c/kmers.c:251-253,278-290); pairing lists of different k is refused."""
    from trio_binning_amd import kmers

    a = kmers.HashSet.from_file(_write(tmp_path, "a.txt", "ACGTACGTACGT\nTTTTTTTTTTTT\nCCCCCCCCCCCA\nGGGGGGGGGGGA\n"))
    b = kmers.HashSet.from_file(_write(tmp_path, "b.txt", "ACGTACGTAC\nAAAAAAAAAA\nCCCCCCCCCA\nGGGGGGGGGA\n"))
    with pytest.raises(ValueError):
        kmers.Classifier(a, b)
    with pytest.raises(ValueError):
        kmers.count_kmers_in_read("ACGTACGTACGTAAAA", a, b)


def test_table_membership_and_dedupe(gpu):
    from trio_binning_amd import kmers

    rng = np.random.default_rng(11)
    keys = rng.integers(0, 2**42, 200_000, dtype=np.uint64)
    keys[1000:2000] = keys[:1000]  # duplicates
    t = kmers.HashSet.from_keys(keys, 21)
    assert t.num_kmers == keys.size
    assert t.distinct == np.unique(keys).size
    assert t.contains(keys).all()
    absent = rng.integers(2**42, 2**43, 50_000, dtype=np.uint64)
    assert not t.contains(absent).any()
    # a crowded table: every line near full, lookups must walk
    t2 = kmers.HashSet.from_keys(keys, 21)
    os.environ["TBK_TABLE_LOAD"] = "0.85"  # read when the list is hashed (first distinct/contains)
    try:
        assert t2.distinct == t.distinct
    finally:
        del os.environ["TBK_TABLE_LOAD"]
    assert t2.contains(keys).all() and not t2.contains(absent).any()
    assert t2.nbytes < t.nbytes
    os.environ["TBK_SHORT"] = "0"   # the key layout stores list lines verbatim, as the reference does (c/kmers.c:113)
    try:
        with kmers.Classifier(t, t2) as cls:
            st = cls.stats()
    finally:
        del os.environ["TBK_SHORT"]
    # the same list on both sides: hapA holds every key, so none is stored for hapB
    assert st["distinct_a"] == t.distinct and st["distinct_b"] == 0 and st["shared_keys"] == keys.size
    assert st["table_bytes"] == st["n_buckets"] * 128
    # short keys (what lists of this size get by themselves) leave out the lines that are not canonical - no window ever
    # asks for them (c/kmers.c:255 looks up min(fwd, rc)) - and again nothing is stored for hapB
    with kmers.Classifier(t, t2) as cls:
        st = cls.stats()
    assert st["short_keys"] and 0.4 * t.distinct < st["distinct_a"] < 0.6 * t.distinct and st["distinct_b"] == 0, st
    assert st["shared_keys"] >= st["distinct_a"] and st["table_bytes"] > st["n_buckets"] * 128, st


@pytest.mark.parametrize("k", [21, 31, 32])
def test_crowded_tables_walk_path(gpu, orc, tmp_path, k, monkeypatch):
    """Crowded layouts (load 0.9 and 0.5): most halves are full, keys go past them to their
    second-choice bucket and on from there, and lookups follow the same sequence."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(21 + k)
    la = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(3000)]
    lb = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(1500)]  # uneven: hapA crowds, hapB has room
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    fb = _write(tmp_path, "b.txt", "".join(x + "\n" for x in lb))
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    reads = _rand_reads(rng, 400, 4000, la + lb, k, p_plant=1.0)
    bases, offs = _pack(reads)
    want = orc.count_batch(bases, offs, oa, ob)
    for load in ("0.9", "0.5"):
        monkeypatch.setenv("TBK_TABLE_LOAD", load)
        # with and without guests: a key whose half is full goes, tagged, into the other list's half of its
        # line before it leaves the line (k < 32; hapB's half has room here), or straight on along its sequence
        for guests in ("1", "0"):
            monkeypatch.setenv("TBK_GUESTS", guests)
            with kmers.Classifier(a, b) as cls:
                assert cls.stats()["n_buckets"] <= 3000 / 8 / float(load) + 1
                got = cls.classify_batch(bases, offs)
            assert np.array_equal(got, want), (k, load, guests, np.nonzero((got != want).any(axis=1))[0][:10])
    assert want[:, 0].sum() > 300 and want[:, 1].sum() > 100


def test_heavy_minimizer_does_not_pile_up(gpu, orc, tmp_path):
    """Low-complexity sequence: thousands of distinct k-mers share one minimizer (a poly-A
    m-mer), far more than a bucket half holds.  The surplus goes to second-choice buckets picked
    by a hash of the whole key, so lookups stay exact and a poly-A-rich read does not walk
    through hundreds of lines per window."""
    import time

    from trio_binning_amd import kmers

    k = 21
    rng = np.random.default_rng(5)
    def around_poly_a(n):
        out = set()
        while len(out) < n:
            left = int(rng.integers(0, 6))
            flank = "".join("ACGT"[c] for c in rng.integers(0, 4, 5))
            out.add(flank[:left] + "A" * 16 + flank[left:])
        return sorted(out)
    pool = around_poly_a(4000)  # 4864 such 21-mers exist
    rng.shuffle(pool)
    la, lb = pool[:2500], pool[2500:]
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    fb = _write(tmp_path, "b.txt", "".join(x + "\n" for x in lb))
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    reads = []
    for _ in range(300):  # poly-A stretches with random interruptions, list k-mers planted, both strands
        parts = []
        for _ in range(60):
            r = rng.random()
            parts.append(pool[int(rng.integers(0, len(pool)))] if r < 0.4 else "A" * int(rng.integers(10, 40)) if r < 0.8
                         else "".join("ACGT"[c] for c in rng.integers(0, 4, 7)))
        s = "".join(parts)
        reads.append(s if rng.random() < 0.5 else _rc(s))
    bases, offs = _pack(reads)
    want = orc.count_batch(bases, offs, oa, ob)
    with kmers.Classifier(a, b) as cls:
        got = cls.classify_batch(bases, offs)
        t0 = time.perf_counter()
        for _ in range(5):
            cls.classify_batch(bases, offs)
        dt = (time.perf_counter() - t0) / 5
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:10]
    assert want[:, 0].sum() > 2500 and want[:, 1].sum() > 1500
    assert dt < 0.5, dt  # ~0.5 Mbases: milliseconds when the surplus is scattered


@pytest.mark.parametrize("load", ["0.1", "0.9"])
def test_shared_keys_count_for_hap_a_only(gpu, orc, tmp_path, load, monkeypatch):
    """hapA is asked first (c/kmers.c:291-294): a key both lists hold counts for hapA only.  The
    table leaves such keys out of hapB's half; counts must match the reference's on lists that
    overlap by half and on disjoint ones, in roomy and crowded tables."""
    from trio_binning_amd import kmers

    k = 21
    rng = np.random.default_rng(99)
    pool = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(4500)]
    la, lb_shared, lb_disjoint = pool[:3000], pool[1500:4500] + pool[1500:1510], pool[3000:4500] + pool[3000:3010]
    monkeypatch.setenv("TBK_TABLE_LOAD", load)
    reads = _rand_reads(rng, 300, 4000, pool, k, p_plant=1.0)
    bases, offs = _pack(reads)
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    oa, a = orc.table_from_file(fa), kmers.HashSet.from_file(fa)
    for name, lb, want_shared, want_b in (("shared", lb_shared, 1510, 1500), ("disjoint", lb_disjoint, 0, 1500)):
        fb = _write(tmp_path, name + ".txt", "".join(x + "\n" for x in lb))
        ob, b = orc.table_from_file(fb), kmers.HashSet.from_file(fb)
        assert b.num_kmers == len(lb)  # the score's denominator still counts every line
        with kmers.Classifier(a, b) as cls:
            st = cls.stats()
            assert (st["shared_keys"], st["distinct_a"], st["distinct_b"]) == (want_shared, 3000, want_b)
            got = cls.classify_batch(bases, offs)
        want = orc.count_batch(bases, offs, oa, ob)
        assert np.array_equal(got, want), (name, load, np.nonzero((got != want).any(axis=1))[0][:10])
        assert want[:, 0].sum() > 300 and want[:, 1].sum() > 100


def test_streaming_order_and_overlap(gpu, orc, tmp_path):
    """submit/wait with several batches in flight returns each batch's own counts."""
    from trio_binning_amd import _lib, kmers

    fa, fb = os.path.join(DATA, "hapA.txt"), os.path.join(DATA, "hapB.txt")
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    lists = [l.strip() for l in open(fa)] + [l.strip() for l in open(fb)]
    rng = np.random.default_rng(4)
    batches = [_rand_reads(rng, int(rng.integers(1, 400)), 2500, lists, 21) for _ in range(7)]
    with kmers.Classifier(a, b) as cls:
        assert cls.depth >= 2
        pending, results = [], []
        for reads in batches:
            if len(pending) == cls.depth:
                results.append(cls.wait(pending.pop(0)))
            pending.append(cls.submit(*_pack(reads)))
        with pytest.raises(_lib.TbkError):  # ring full
            for _ in range(cls.depth + 1):
                pending.append(cls.submit(*_pack(batches[0])))
        pending = pending[:cls.depth]
        while pending:
            results.append(cls.wait(pending.pop(0)))
    for reads, got in zip(batches, results[:len(batches)]):
        bases, offs = _pack(reads)
        assert np.array_equal(got, orc.count_batch(bases, offs, oa, ob))


def test_lists_choose_the_line_layout(gpu, orc, monkeypatch):
    """Without TBK_FRONT the lists decide in which layout the probe reads a bucket's line: keys that fall
    evenly into buckets are read front-first (one build), lists that cluster the way real find-unique-kmers
    output does (runs of overlapping k-mers around variants) are built again in whole lines - same sampling
    rule (mod-sampling unless TBK_MOD_SAMPLING=0), same load (0.08 unless TBK_TABLE_LOAD).  Either way, and
    with the rule, the layout or the load pinned, the counts are the oracle's."""
    import ctypes as C

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    monkeypatch.delenv("TBK_MOD_SAMPLING", raising=False)
    monkeypatch.delenv("TBK_TABLE_LOAD", raising=False)
    monkeypatch.delenv("TBK_FRONT", raising=False)
    monkeypatch.setenv("TBK_SHORT", "0")   # the KEY layouts' policy (lists of this size that do not merge get short keys first: tests/test_gpu_entry.py)
    dev, k, n = 0, 21, 400_000

    def dalloc(nbytes):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, nbytes, C.byref(p)))
        return p.value

    # uniform: BASELINE's synthetic lists; clustered: haplotype-shaped lists of a 30 Mb genome
    uni = np.empty(2 * n, dtype=np.uint64)
    check(lib.tbk_synth_keys_host(0x5EED0001, 0, 2 * n, k, uni.ctypes.data))
    cap = 2 * n
    d_a, d_b = dalloc(cap * 8), dalloc(cap * 8)
    got = C.c_uint64()
    check(lib.tbk_synth_hap_keys_device(dev, 0x5EED0001, 3_000_000, int(round((1 / 500) * (1 << 24))), k, C.c_void_p(d_a), C.c_void_p(d_b), cap, C.byref(got)))
    m = got.value
    assert 100_000 < m <= cap
    hap = np.empty(2 * m, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(dev, hap.ctypes.data, C.c_void_p(d_a), m * 8))
    check(lib.tbk_memcpy_d2h(dev, hap.ctypes.data + m * 8, C.c_void_p(d_b), m * 8))
    for p in (d_a, d_b):
        check(lib.tbk_device_free(dev, C.c_void_p(p)))
    rng = np.random.default_rng(8)

    def decode(key):
        return "".join("ACGT"[(int(key) >> (2 * i)) & 3] for i in range(k))

    for name, keys, half, want_front, want_builds in (("uniform", uni, n, True, 1), ("clustered", hap, m, False, 2)):
        ka, kb = keys[:half], keys[half:]
        oa, ob = orc.table_from_keys(ka, k), orc.table_from_keys(kb, k)
        plants = [decode(x) for x in np.concatenate([ka[:300], kb[:300]])]
        reads = _rand_reads(rng, 300, 3000, plants, k, p_plant=0.9)
        bases, offs = _pack(reads)
        want = orc.count_batch(bases, offs, oa, ob)
        assert want.sum() > 300
        a, b = kmers.HashSet.from_keys(ka, k), kmers.HashSet.from_keys(kb, k)
        with kmers.Classifier(a, b) as cls:
            st = cls.stats()
            assert st["sampling_t"] > 0 and st["layout_builds"] in (want_builds, want_builds + (name == "clustered")), (name, st)
            # lists that spread are probed front-first (64 of a line's 128 bytes; few of their keys lie behind the first
            # four slots of a bucket); clustered ones - runs of overlapping k-mers - are rebuilt as entries (k = 21 has
            # room for them: tests/test_gpu_entry.py), which takes them out of the key layouts altogether
            assert st["front_layout"] == want_front and (not want_front or st["keys_behind_front"] <= 0.006 * 2 * half), (name, st)
            assert st["entry_layout"] == (name == "clustered"), (name, st)
            load = half / (st["n_buckets"] * 8)
            assert st["entry_layout"] or abs(load - 0.08) < 0.005, (name, load)   # 100 B of HBM per key in the key layouts
            assert np.array_equal(cls.classify_batch(bases, offs), want), name
        for pin in ("0", "1"):
            monkeypatch.setenv("TBK_MOD_SAMPLING", pin)   # the rule is pinned, the layout still follows the lists
            with kmers.Classifier(a, b) as cls:
                st = cls.stats()
                assert (st["sampling_t"] > 0) == (pin == "1") and st["layout_builds"] >= want_builds and st["front_layout"] == want_front, (name, pin, st)
                assert st["entry_layout"] == (name == "clustered" and pin == "1"), (name, pin, st)   # (entries need mod-sampling's position; the random minimizer keeps whole lines)
                assert np.array_equal(cls.classify_batch(bases, offs), want), (name, pin)
            monkeypatch.setenv("TBK_TABLE_LOAD", "0.1")   # and the load
            with kmers.Classifier(a, b) as cls:
                st = cls.stats()
                assert (st["sampling_t"] > 0) == (pin == "1") and st["front_layout"] == want_front, (name, pin, st)
                assert st["entry_layout"] or abs(half / (st["n_buckets"] * 8) - 0.1) < 0.005
                assert np.array_equal(cls.classify_batch(bases, offs), want), (name, pin, "load")
            monkeypatch.delenv("TBK_TABLE_LOAD")
        monkeypatch.delenv("TBK_MOD_SAMPLING")
        for front in ("0", "1"):                          # the layout pinned either way: one build
            monkeypatch.setenv("TBK_FRONT", front)
            with kmers.Classifier(a, b) as cls:
                st = cls.stats()
                assert st["front_layout"] == (front == "1") and st["layout_builds"] == 1 and st["sampling_t"] > 0, (name, front, st)
                assert np.array_equal(cls.classify_batch(bases, offs), want), (name, "front", front)
        monkeypatch.delenv("TBK_FRONT")


@pytest.mark.parametrize("k,front", [(21, 1), (21, 0), (31, 1), (16, 1)])
def test_two_read_passes_at_every_boundary_offset(gpu, orc, monkeypatch, k, front):
    """Passes that touch exactly two reads run through a kernel of their own: the boundary is folded into the lanes'
    masks and into scalar ownership masks.  Here the boundary falls on every offset inside a lane (0..31), next to
    the pass's first and last window, and the bases on both sides of it are dense with list k-mers - including
    k-mers that would match ACROSS the boundary if the two reads were one.  Counts are the oracle's, with the
    kernel (default) and without it (TBK_TWO_READ is read once per process: compared through the sliced and
    unsliced paths instead)."""
    from trio_binning_amd import kmers

    for v in ("TBK_MINIMIZER_W", "TBK_MINIMIZER_M", "TBK_MOD_SAMPLING", "TBK_TABLE_LOAD"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("TBK_FRONT", str(front))
    rng = np.random.default_rng(100 * k + front)
    genome = "".join("ACGT"[c] for c in rng.integers(0, 4, 400_000))
    # every other k-mer of the genome is in a list: reads cut from it hit in nearly every window
    starts = rng.permutation(len(genome) - k)[:60_000]
    la = [genome[p:p + k] for p in starts[:30_000]]
    lb = [genome[p:p + k] for p in starts[30_000:]]
    canon = lambda x: min(x, _rc(x))
    ka = np.array([kmers.kmer_to_int(canon(x)) for x in la], dtype=np.uint64)
    kb = np.array([kmers.kmer_to_int(canon(x)) for x in lb], dtype=np.uint64)
    oa, ob = orc.table_from_keys(ka, k), orc.table_from_keys(kb, k)
    a, b = kmers.HashSet.from_keys(ka, k), kmers.HashSet.from_keys(kb, k)
    # consecutive pieces of the genome: cut where one read ends the next begins, so a window over the cut would be a
    # genome k-mer (a hit) if the kernel let it through.  Piece lengths put the cuts on every offset mod 32, at
    # distance 1 .. k from both ends of a pass, and leave passes with exactly two reads between longer reads.
    lengths = []
    for d in list(range(0, 34)) + [2048 - 1, 2048 - k, 2048 - k + 1, 1, k - 1, k, k + 1]:
        lengths += [2048 + 700 + d, 2048 * 2 - 700]
    lengths += [15000] * 6 + [2048] * 3 + [4096 + 5, 2043, 33, 2048 * 3 - 38]
    reads, at = [], 0
    for n in lengths:
        if at + n > len(genome):
            at = int(rng.integers(0, 1000))
        reads.append(genome[at:at + n])
        at += n
    for slice_bases in ("2048", str(1 << 30)):   # an empty ring takes the batch slice by slice / in one piece
        monkeypatch.setenv("TBK_SLICE_BASES", slice_bases)
        for order in (reads, reads[::-1]):
            bases, offs = _pack(order)
            want = orc.count_batch(bases, offs, oa, ob)
            assert want.sum() > len(bases) // 8
            with kmers.Classifier(a, b) as cls:
                st = cls.stats()
                assert st["front_layout"] == bool(front), st
                got = cls.classify_batch(bases, offs)
                assert np.array_equal(got, want), (k, front, slice_bases, np.nonzero((got != want).any(axis=1))[0][:10])
                n_passes, n_listed = cls.last_passes()
                assert n_listed >= 40   # most cuts leave a pass with two reads


@pytest.mark.parametrize("k,want_w", [(21, 7), (22, 7), (23, 8), (25, 8), (31, 8), (32, 7)])  # (lists this small leave room for m = 15 or 16; 2 x 3e8 21-mers: w = 6, m = 16)
def test_span_follows_k(gpu, orc, monkeypatch, k, want_w):
    """Without TBK_MINIMIZER_W the span is as long as k leaves room for, up to 8 m-mers (front layout: fewer line
    switches per window); lists that cluster fall back to whole lines and a span of at most 6.  Counts are the
    oracle's either way."""
    import ctypes as C

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    for v in ("TBK_MINIMIZER_W", "TBK_MINIMIZER_M", "TBK_MOD_SAMPLING", "TBK_TABLE_LOAD", "TBK_FRONT"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("TBK_SHORT", "0")   # (the key layouts' spans; short keys take the front layout's span: tests/test_gpu_entry.py)
    n = 200_000
    uni = np.empty(2 * n, dtype=np.uint64)
    check(lib.tbk_synth_keys_host(0x5EED0001, 0, 2 * n, k, uni.ctypes.data))
    ka, kb = uni[:n], uni[n:]
    oa, ob = orc.table_from_keys(ka, k), orc.table_from_keys(kb, k)
    rng = np.random.default_rng(k)

    def decode(key):
        return "".join("ACGT"[(int(key) >> (2 * i)) & 3] for i in range(k))

    plants = [decode(x) for x in np.concatenate([ka[:200], kb[:200]])]
    reads = _rand_reads(rng, 200, 3000, plants, k, p_plant=0.9)
    bases, offs = _pack(reads)
    want = orc.count_batch(bases, offs, oa, ob)
    assert want.sum() > 200
    a, b = kmers.HashSet.from_keys(ka, k), kmers.HashSet.from_keys(kb, k)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["minimizer_w"] == want_w and st["front_layout"] and st["sampling_t"] > 0 and st["layout_builds"] == 1, st
        span = st["minimizer_m"] + st["minimizer_w"] - 1
        assert span <= k and (k - span) % 2 == 0, st
        assert np.array_equal(cls.classify_batch(bases, offs), want)
    monkeypatch.setenv("TBK_FRONT", "0")  # whole lines (what clustered lists get): the span stays at 6 or below
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["minimizer_w"] <= 6 and not st["front_layout"], st
        assert np.array_equal(cls.classify_batch(bases, offs), want)


def test_prepacked_batches(gpu, orc, transfer):
    """tbk_pack_bases + tbk_stream_submit_packed: batches packed ahead of time (pinned and pageable
    arrays), with N runs, lower case, reads ending inside a chunk, an empty batch; counts equal the
    oracle's and the ASCII path's.  A malformed exception list is refused."""
    import ctypes as C

    from trio_binning_amd import _lib, kmers
    from trio_binning_amd._lib import lib

    fa, fb = os.path.join(DATA, "hapA.txt"), os.path.join(DATA, "hapB.txt")
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    lists = [l.strip() for l in open(fa)] + [l.strip() for l in open(fb)]
    rng = np.random.default_rng(44)
    with kmers.Classifier(a, b) as cls:
        assert cls.packed_transfer == (transfer == "packed-h2d")
        for trial in range(6):
            reads = _rand_reads(rng, int(rng.integers(1, 300)), 3000, lists, 21)
            for i in range(0, len(reads), 5):   # damage some reads: N runs, lower case, a planted k-mer broken by an N
                r = list(reads[i])
                if len(r) > 60:
                    p = int(rng.integers(0, len(r) - 40))
                    r[p:p + int(rng.integers(1, 40))] = "N" * 1
                    r[-3:] = "acg"
                reads[i] = "".join(r)
            if trial == 3:
                reads = ["", "ACGT", lists[0], lists[0] + "N" + lists[1], "N" * 50, lists[2][:20]]
            bases, offs = _pack(reads)
            want = orc.count_batch(bases, offs, oa, ob)
            for pinned in (True, False):
                pk = kmers.pack_bases(bases, offs, pinned=pinned)
                assert np.array_equal(cls.wait(cls.submit_packed(pk)), want), (trial, pinned)
            assert np.array_equal(cls.classify_batch(bases, offs), want)
            cls.packed_transfer = not cls.packed_transfer      # and through the other transfer of submit
            assert np.array_equal(cls.classify_batch(bases, offs), want)
            cls.packed_transfer = not cls.packed_transfer
        empty = kmers.pack_bases(np.zeros(0, dtype=np.uint8), np.array([0, 0, 0], dtype=np.uint64))
        assert cls.wait(cls.submit_packed(empty)).tolist() == [[0, 0], [0, 0]]
        pk = kmers.pack_bases(*_pack([lists[0] * 3]), pinned=False)
        bad_chunk = np.array([99], dtype=np.uint32)
        counts, tk = np.zeros((1, 2), dtype=np.int32), C.c_uint64()
        assert lib.tbk_stream_submit_packed(cls._h, pk.codes.ctypes.data, bad_chunk.ctypes.data, pk.exc_mask.ctypes.data, 1,
                                            pk.offsets.ctypes.data, 1, counts.ctypes.data, C.byref(tk)) == _lib.TBK_ERR_INVALID
        assert lib.tbk_stream_submit_packed(cls._h, pk.codes.ctypes.data, None, None, 1,
                                            pk.offsets.ctypes.data, 1, counts.ctypes.data, C.byref(tk)) == _lib.TBK_ERR_INVALID
        assert np.array_equal(cls.wait(cls.submit_packed(pk)), orc.count_batch(*_pack([lists[0] * 3]), oa, ob))


def test_realistic_haplotypes(gpu, orc):
    """Lists shaped like real trio-binning input rather than uniform random keys: two haplotypes
    of one 2 Mb genome differing by SNPs, each list = the canonical 21-mers found in one
    haplotype only (so keys come in runs of up to 21 overlapping k-mers that share minimizers),
    reads sampled from either haplotype on either strand with sequencing errors and a few N.
    Dense hits, clustered buckets, both lists hit inside one read."""
    from trio_binning_amd import kmers

    k, glen = 21, 2_000_000
    rng = np.random.default_rng(2024)
    genome = rng.integers(0, 4, glen, dtype=np.uint8)

    def mutate(g, rate, seed):
        r = np.random.default_rng(seed)
        h = g.copy()
        pos = np.nonzero(r.random(g.size) < rate)[0]
        h[pos] = (h[pos] + r.integers(1, 4, pos.size)) % 4
        return h

    hap = [mutate(genome, 1 / 500, 1), mutate(genome, 1 / 500, 2)]

    def canon_kmers(codes):
        c = codes.astype(np.uint64)
        n = c.size - k + 1
        fwd = np.zeros(n, dtype=np.uint64)
        rc = np.zeros(n, dtype=np.uint64)
        for i in range(k):
            fwd |= c[i:i + n] << np.uint64(2 * i)
            rc |= (np.uint64(3) - c[i:i + n]) << np.uint64(2 * (k - 1 - i))
        return np.minimum(fwd, rc)

    ka, kb = np.unique(canon_kmers(hap[0])), np.unique(canon_kmers(hap[1]))
    only_a, only_b = np.setdiff1d(ka, kb), np.setdiff1d(kb, ka)
    assert only_a.size > 50_000 and only_b.size > 50_000
    a, b = kmers.HashSet.from_keys(only_a, k), kmers.HashSet.from_keys(only_b, k)
    oa, ob = orc.table_from_keys(only_a, k), orc.table_from_keys(only_b, k)

    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    reads = []
    for i in range(600):
        h = hap[i % 2]
        n = int(rng.integers(200, 12_000))
        p = int(rng.integers(0, glen - n))
        codes = h[p:p + n].copy()
        if rng.random() < 0.5:
            codes = comp[codes[::-1]]
        err = rng.random(n) < 0.005
        codes[err] = (codes[err] + rng.integers(1, 4, int(err.sum()))) % 4
        s = lut[codes].copy()
        if i % 7 == 0:
            s[rng.integers(0, n, 3)] = ord("N")
        reads.append(s.tobytes().decode())
    bases, offs = _pack(reads)
    want = orc.count_batch(bases, offs, oa, ob)
    for load in ("0.125", "0.5"):  # roomy and crowded layouts of the same clustered keys
        os.environ["TBK_TABLE_LOAD"] = load
        try:
            cls = kmers.Classifier(a, b)
        finally:
            del os.environ["TBK_TABLE_LOAD"]
        with cls:
            assert cls.stats()["shared_keys"] == 0
            got = cls.classify_batch(bases, offs)
        assert np.array_equal(got, want), (load, np.nonzero((got != want).any(axis=1))[0][:10])
    # reads from haplotype A carry mostly hapA k-mers and vice versa
    assert (want[0::2, 0] > want[0::2, 1]).mean() > 0.95 and (want[1::2, 1] > want[1::2, 0]).mean() > 0.95
    assert want.sum() > 100_000


def test_device_resident_and_synthetic_generators(gpu, orc):
    """The bench path at test size: keys and reads generated on the GPU, classified from HBM
    through the ticket ring into pinned host memory; checked against the oracle on the same
    bytes copied back, plus the generator's own promises (distinct canonical keys, host and
    device sequences identical, planted k-mers found)."""
    import ctypes as C

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    dev, k, n_list, R, L = 0, 21, 20000, 300, 3000
    seed_k, seed_r = 0x5EED0001, 0x5EED0002

    def dalloc(n):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, n, C.byref(p)))
        return p.value

    d_keys = dalloc(2 * n_list * 8)
    check(lib.tbk_synth_keys_device(dev, seed_k, 0, 2 * n_list, k, C.c_void_p(d_keys)))
    h_keys = np.empty(2 * n_list, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(dev, h_keys.ctypes.data, C.c_void_p(d_keys), h_keys.nbytes))
    host_keys = np.empty(2 * n_list, dtype=np.uint64)
    check(lib.tbk_synth_keys_host(seed_k, 0, 2 * n_list, k, host_keys.ctypes.data))
    assert np.array_equal(h_keys, host_keys)
    assert np.unique(h_keys).size == 2 * n_list
    for key in h_keys[:200]:  # canonical: packed value <= its reverse complement's
        s = "".join("ACGT"[(int(key) >> (2 * i)) & 3] for i in range(k))
        assert orc.kmer_to_int(s) == int(key) and int(key) <= orc.kmer_to_int(_rc(s))
    a = kmers.HashSet.from_device_keys(d_keys, n_list, k)
    b = kmers.HashSet.from_device_keys(d_keys + n_list * 8, n_list, k)
    check(lib.tbk_device_free(dev, C.c_void_p(d_keys)))
    oa, ob = orc.table_from_keys(h_keys[:n_list], k), orc.table_from_keys(h_keys[n_list:], k)

    total = R * L
    d_bases, d_offs = dalloc(total + 32), dalloc((R + 1) * 8)
    check(lib.tbk_synth_reads_device(dev, seed_r, 0, R, L, seed_k, n_list, n_list, k, 30, 3, C.c_void_p(d_bases), C.c_void_p(d_offs)))
    h_bases = np.empty(total, dtype=np.uint8)
    h_offs = np.empty(R + 1, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(dev, h_bases.ctypes.data, C.c_void_p(d_bases), total))
    check(lib.tbk_memcpy_d2h(dev, h_offs.ctypes.data, C.c_void_p(d_offs), h_offs.nbytes))
    assert h_offs.tolist() == [i * L for i in range(R + 1)]
    assert set(np.unique(h_bases).tolist()) <= {65, 67, 71, 84}
    want = orc.count_batch(h_bases, h_offs, oa, ob)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["distinct_a"] == n_list and st["distinct_b"] == n_list
        pinned = [kmers.pinned_empty((R, 2), np.int32) for _ in range(2)]
        t0 = cls.submit_device(d_bases, d_offs, R, total, pinned[0])
        t1 = cls.submit_device(d_bases, d_offs, R, total, pinned[1])
        plain = np.zeros((R, 2), dtype=np.int32)  # unpinned destination: staged inside the library
        t2 = cls.submit_device(d_bases, d_offs, R, total, plain)
        for t in (t0, t1, t2):
            cls.wait(t)
        d_counts = dalloc(R * 8)
        cls.classify_device(d_bases, d_offs, R, total, d_counts)
        cls.sync()
        direct = np.zeros((R, 2), dtype=np.int32)
        check(lib.tbk_memcpy_d2h(dev, direct.ctypes.data, C.c_void_p(d_counts), direct.nbytes))
    for got in (pinned[0], pinned[1], plain, direct):
        assert np.array_equal(got, want)
    # planted: origin reads carry >= 30 k-mers of their list, 3 of the other; the rest 3 + 3
    major = want.max(axis=1)
    assert ((major >= 30) | (major <= 6)).all() and (major >= 30).sum() > R // 2
    for p in (d_bases, d_offs, d_counts):
        check(lib.tbk_device_free(dev, C.c_void_p(p)))


def test_haplotype_shaped_generators(gpu, orc):
    """bench.py --lists haplotypes at test size: the GPU generators against a numpy restatement
    (same hash, same rules), then the probe kernel on what they made against the oracle."""
    import ctypes as C

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    dev, k, G, R, L = 0, 21, 300_000, 200, 2500
    seed, rseed, snp24, err24 = 0x5EED0001, 0x5EED0002, int((1 / 400) * (1 << 24)), int(0.003 * (1 << 24))
    M = np.uint64

    def splitmix(x):
        with np.errstate(over="ignore"):
            x = x + M(0x9E3779B97F4A7C15)
            x = (x ^ (x >> M(30))) * M(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> M(27))) * M(0x94D049BB133111EB)
            return x ^ (x >> M(31))

    with np.errstate(over="ignore"):
        r = splitmix(M(seed) ^ (np.arange(G, dtype=M) * M(0x9E3779B97F4A7C15)))
    base = (r & M(3)).astype(np.uint8)
    alt_a = (base + 1 + (((r >> M(52)) & M(0xFF)) % M(3)).astype(np.uint8)) & 3
    alt_b = (base + 1 + (((r >> M(56)) & M(0xFF)) % M(3)).astype(np.uint8)) & 3
    hap_a = np.where(((r >> M(4)) & M(0xFFFFFF)) < snp24, alt_a, base).astype(np.uint8)
    hap_b = np.where(((r >> M(28)) & M(0xFFFFFF)) < snp24, alt_b, base).astype(np.uint8)

    def canon(codes):
        c = codes.astype(M)
        n = c.size - k + 1
        f, rc = np.zeros(n, dtype=M), np.zeros(n, dtype=M)
        for i in range(k):
            f |= c[i:i + n] << M(2 * i)
            rc |= (M(3) - c[i:i + n]) << M(2 * (k - 1 - i))
        return np.minimum(f, rc)

    diff = np.concatenate([[0], np.cumsum(hap_a != hap_b)])
    covers = (diff[k:] - diff[:-k]) > 0  # window q covers a differing position
    want_a, want_b = canon(hap_a)[covers], canon(hap_b)[covers]

    def dalloc(n):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, n, C.byref(p)))
        return p.value

    cap = int(covers.sum()) + 100
    d_keys = dalloc(2 * cap * 8)
    n_got = C.c_uint64()
    check(lib.tbk_synth_hap_keys_device(dev, seed, G, snp24, k, C.c_void_p(d_keys), C.c_void_p(d_keys + cap * 8), cap, C.byref(n_got)))
    n = n_got.value
    assert n == int(covers.sum()) and n > 10_000
    got = np.empty(2 * cap, dtype=M)
    check(lib.tbk_memcpy_d2h(dev, got.ctypes.data, C.c_void_p(d_keys), got.nbytes))
    ga, gb = got[:n], got[cap:cap + n]
    # order is not deterministic; the multiset of (A key, B key) pairs is
    assert np.array_equal(np.sort(ga), np.sort(want_a)) and np.array_equal(np.sort(gb), np.sort(want_b))
    order_g, order_w = np.lexsort((gb, ga)), np.lexsort((want_b, want_a))
    assert np.array_equal(gb[order_g], want_b[order_w])

    total = R * L
    d_bases, d_offs = dalloc(total + 32), dalloc((R + 1) * 8)
    check(lib.tbk_synth_hap_reads_device(dev, seed, G, snp24, rseed, 7, R, L, err24, C.c_void_p(d_bases), C.c_void_p(d_offs)))
    h_bases, h_offs = np.empty(total, dtype=np.uint8), np.empty(R + 1, dtype=M)
    check(lib.tbk_memcpy_d2h(dev, h_bases.ctypes.data, C.c_void_p(d_bases), total))
    check(lib.tbk_memcpy_d2h(dev, h_offs.ctypes.data, C.c_void_p(d_offs), h_offs.nbytes))
    assert h_offs.tolist() == [i * L for i in range(R + 1)]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with np.errstate(over="ignore"):
        rr = splitmix(M(rseed) ^ ((M(7) + np.arange(R, dtype=M)) * M(0xA0761D6478BD642F)))
        e = splitmix(M(rseed) ^ M(0x5851F42D4C957F2D) ^ ((M(7 * L) + np.arange(total, dtype=M)) * M(0xD6E8FEB86659FD93)))
    want_reads = np.empty((R, L), dtype=np.uint8)
    for i in range(R):
        start = int((rr[i] >> M(1)) % M(G - L + 1))
        codes = (hap_b if (7 + i) & 1 else hap_a)[start:start + L]
        want_reads[i] = (3 - codes[::-1]) if int(rr[i]) & 1 else codes
    want_reads = want_reads.reshape(-1)
    hit = (e & M(0xFFFFFF)) < err24
    want_reads = np.where(hit, (want_reads + 1 + ((e >> M(32)) % M(3)).astype(np.uint8)) & 3, want_reads).astype(np.uint8)
    assert np.array_equal(h_bases, lut[want_reads])

    a, b = kmers.HashSet.from_device_keys(d_keys, n, k), kmers.HashSet.from_device_keys(d_keys + cap * 8, n, k)
    oa, ob = orc.table_from_keys(ga, k), orc.table_from_keys(gb, k)
    want = orc.count_batch(h_bases, h_offs, oa, ob)
    with kmers.Classifier(a, b) as cls:
        got_counts = cls.classify_batch(h_bases, h_offs)
    assert np.array_equal(got_counts, want)
    # reads of haplotype A (even read index 7 + i) carry mostly hapA k-mers
    even = (7 + np.arange(R)) % 2 == 0
    assert (want[even, 0] > want[even, 1]).mean() > 0.9 and (want[~even, 1] > want[~even, 0]).mean() > 0.9
    for p in (d_keys, d_bases, d_offs):
        check(lib.tbk_device_free(dev, C.c_void_p(p)))


def test_score_and_bin_on_gpu_counts(gpu, orc):
    from trio_binning_amd import kmers

    counts = np.array([[4, 1], [0, 2], [0, 0], [3, 4], [7, 7]], dtype=np.int32)
    sa, sb, bins = kmers.score_and_bin(counts, 4, 3)
    osa, osb, obins = orc.score_and_bin(counts, 4, 3)
    assert sa.tolist() == osa.tolist() and sb.tolist() == osb.tolist() and bins.decode() == obins
    assert bins == b"ABUBB"


def test_abi_misuse_is_reported_not_crashed(gpu, tmp_path):
    """Error convention of the boundary: bad arguments come back as status + message
    (ValueError / TbkError in Python), never as a crash or a silent wrong answer."""
    import ctypes as C

    from trio_binning_amd import _lib, kmers
    from trio_binning_amd._lib import lib

    a = kmers.HashSet.from_file(os.path.join(DATA, "hapA.txt"))
    b = kmers.HashSet.from_file(os.path.join(DATA, "hapB.txt"))
    with kmers.Classifier(a, b) as cls:
        bases = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTACGT", dtype=np.uint8)
        counts = np.zeros((1, 2), dtype=np.int32)
        tk = C.c_uint64()
        bad_first = np.array([3, 28], dtype=np.uint64)
        assert lib.tbk_stream_submit(cls._h, bases.ctypes.data, bad_first.ctypes.data, 1, counts.ctypes.data, C.byref(tk)) == _lib.TBK_ERR_INVALID
        assert "offsets[0]" in _lib.last_error()
        decreasing = np.array([0, 20, 10], dtype=np.uint64)
        c2 = np.zeros((2, 2), dtype=np.int32)
        assert lib.tbk_stream_submit(cls._h, bases.ctypes.data, decreasing.ctypes.data, 2, c2.ctypes.data, C.byref(tk)) == _lib.TBK_ERR_INVALID
        assert lib.tbk_stream_submit(cls._h, None, None, 1, counts.ctypes.data, C.byref(tk)) == _lib.TBK_ERR_INVALID
        assert lib.tbk_stream_wait(cls._h, 12345) == _lib.TBK_ERR_STATE
        assert lib.tbk_classify_device(cls._h, C.c_void_p(8), C.c_void_p(16), 1, 10, C.c_void_p(16)) == _lib.TBK_ERR_INVALID  # misaligned
        # a good call still works afterwards
        assert cls.classify_reads(["ACCTCTAAGAAGCTTTGAAAA"]).tolist() == [[1, 0]]
        # the counter takes the same batches and applies the same rule to their offsets
        with kmers.KmerCounter(5, 1 << 10) as ctr:
            for bad in (bad_first, decreasing):
                assert lib.tbk_counter_add_batch(ctr._h, bases.ctypes.data, bad.ctypes.data, bad.size - 1) == _lib.TBK_ERR_INVALID
            assert ctr.stats()["bases_added"] == 0
    h = C.c_void_p()
    assert lib.tbk_table_create_from_keys(None, 5, 21, 0, C.byref(h)) == _lib.TBK_ERR_INVALID
    keys = np.arange(4, dtype=np.uint64)
    assert lib.tbk_table_create_from_keys(keys.ctypes.data, 4, 33, 0, C.byref(h)) == _lib.TBK_ERR_INVALID
    assert lib.tbk_table_create_from_keys(keys.ctypes.data, 0, 21, 0, C.byref(h)) == _lib.TBK_ERR_FORMAT
    assert lib.tbk_table_create_from_keys(keys.ctypes.data, 4, 21, 99, C.byref(h)) == _lib.TBK_ERR_INVALID  # no such device
    assert lib.tbk_classifier_create(None, None, C.byref(h)) == _lib.TBK_ERR_INVALID
    lib.tbk_table_destroy(None)
    lib.tbk_classifier_destroy(None)
