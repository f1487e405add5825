"""The entry layout (csrc/tbk_common.h "entry layout"): a run of overlapping list k-mers stored once per sampled
m-mer, compared under the window's mask by a pair-cooperative probe.  Membership must be exactly the reference's
(kmer_in_hash_set, c/kmers.c:245-268; count_kmers_in_read, c/kmers.c:270-299): every test compares the HIP path,
through the C-ABI, with the oracle or with the recorded output of the real reference.  The CPU model of the layout's
arithmetic is tests/test_entry_model.py."""
import os

import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import _pack, _rand_reads, _rc, _write

pytestmark = pytest.mark.gpu


def _genome_lists(rng, k, n_loci, snp_every=150, genome=40_000):
    """Two haplotypes of a random genome that differ by SNPs; a list = the k-mers of one haplotype the other lacks, as
    find-unique-kmers writes them (canonical, find_unique_kmers.py:200-233), in runs of up to k around every SNP."""
    ga = rng.integers(0, 4, genome)
    ga[5000:5300] = (np.arange(300) // 3) & 1               # low complexity
    ga[9000:11000] = ga[2000:4000]                           # a repeat
    pal = rng.integers(0, 4, 40)
    ga[12000:12080] = np.concatenate([pal, 3 - pal[::-1]])   # its own reverse complement: palindromic m-mers and k-mers
    gb = ga.copy()
    at = rng.choice(genome, size=genome // snp_every, replace=False)
    gb[at] = (ga[at] + rng.integers(1, 4, at.size)) & 3
    sa, sb = "".join("ACGT"[c] for c in ga), "".join("ACGT"[c] for c in gb)

    def canon(s):
        r = _rc(s)
        # the reference's canonical form is the smaller PACKED integer (base i at bits 2i): compare from the last base down
        return s if s[::-1] <= r[::-1] else r

    ka = {canon(sa[i:i + k]) for i in range(genome - k + 1)}
    kb = {canon(sb[i:i + k]) for i in range(genome - k + 1)}
    la, lb = sorted(ka - kb), sorted(kb - ka)
    rng.shuffle(la)
    rng.shuffle(lb)
    return sa, sb, la[:n_loci], lb[:n_loci]


def _case(rng, k, tmp_path, orc, crowd_cores=0):
    """Lists, reads and the oracle's counts of one layout test (see test_entry_layout_counts_equal_the_oracle); crowd_cores:
    that many stretches of sequence with 200 list k-mers each that differ from it in their end bases (all of them sample
    an m-mer of the stretch: lines that overflow whatever the table's size)."""
    from trio_binning_amd import kmers

    sa, sb, la, lb = _genome_lists(rng, k, 6000)
    uni = ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(1500)]
    la += uni[:700]
    lb += uni[700:1400]
    lb += [la[int(i)] for i in rng.integers(0, len(la), 40)] + [_rc(la[int(i)]) for i in rng.integers(0, len(la), 40)]   # shared with hapA
    la += [la[int(i)] for i in rng.integers(0, len(la), 20)]                                                               # duplicate lines
    cores = []
    for c in range(crowd_cores):
        core = "".join("ACGT"[x] for x in rng.integers(0, 4, k))
        cores.append(core)
        for i in range(200):
            x = list(core)
            for j in (0, 1, 2, k - 3, k - 2, k - 1):
                if rng.random() < 0.5:
                    x[j] = "ACGT"[int(rng.integers(0, 4))]
            (la if (c + i) & 1 else lb).append("".join(x))
    la += ["A" * k, "T" * k, ("AC" * k)[:k], ("ACGT" * k)[:k]]
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    fb = _write(tmp_path, "b.txt", "\n".join(lb))
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    assert (a.num_kmers, b.num_kmers) == (oa.num_kmers, ob.num_kmers)
    reads = []
    for hap in (sa, sb):
        for _ in range(12):
            lo = int(rng.integers(0, len(hap) - 6000))
            r = list(hap[lo:lo + int(rng.integers(50, 6000))])
            for i in rng.integers(0, len(r), len(r) // 300):
                r[int(i)] = "ACGT"[int(rng.integers(0, 4))]
            r = "".join(r)
            reads.append(r if rng.random() < 0.5 else _rc(r))
    reads += _rand_reads(rng, 40, 2500, la + lb, k, p_plant=0.9)
    reads += ["", "A" * (k - 1), la[0], _rc(lb[0]) * 2, "".join(la[:30]), sa[11990:12100], _rc(sa[11990:12100]), sa[4990:5320]]
    body = sa[20000:26000]
    reads += [body[: 2048 - sum(map(len, reads)) % 2048], body[:2047], body[:2048 + k - 1], body[:4096]]   # pass-boundary shapes
    noisy = list(sb[30000:33000])
    for i in rng.integers(0, 3000, 40):
        noisy[int(i)] = "NnacgtR-"[int(rng.integers(0, 8))]
    reads.append("".join(noisy))
    for core in cores:   # variants of the crowded stretches, most of them NOT in the lists, on either strand
        vs = []
        for _ in range(120):
            x = list(core)
            for j in (0, 1, 2, k - 3, k - 2, k - 1):
                if rng.random() < 0.5:
                    x[j] = "ACGT"[int(rng.integers(0, 4))]
            vs.append("".join(x))
        reads += ["".join(vs[:60]), _rc("".join(vs[60:]))]
    reads = [reads[int(i)] for i in rng.permutation(len(reads))]
    bases, offs = _pack(reads)
    want = orc.count_batch(bases, offs, oa, ob, strict=True)
    assert want.sum() > 2000
    return a, b, bases, offs, want, reads


NARROW = [(21, 6, 0), (21, 5, 0), (21, 4, 0), (22, 6, 0), (23, 6, 0), (23, 5, 0), (24, 5, 0), (25, 4, 0)]
WIDE = [(31, 8, 1), (31, 6, 1), (32, 7, 1), (29, 8, 1), (27, 6, 1), (26, 7, 1), (21, 6, 1), (24, 7, 1)]   # wide entries (16 bytes): k up to 32, and any k when asked for


@pytest.mark.parametrize("k,w,wide", NARROW + WIDE)
@pytest.mark.parametrize("crowded", [0, 1])
def test_entry_layout_counts_equal_the_oracle(gpu, orc, tmp_path, monkeypatch, k, w, wide, crowded):
    """Lists of runs (two haplotypes' unique k-mers), uniform keys, duplicate lines, lines shared between the lists on
    either strand, non-canonical lines (dead in the reference), low-complexity and palindromic sequence; reads drawn
    from both haplotypes on both strands with errors, ragged shapes, bytes outside ACGT - in a roomy table and in one
    so crowded that lists overflow their lines (second looks, walks)."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(100 * k + 10 * w + crowded)
    monkeypatch.setenv("TBK_ENTRY", "1")
    monkeypatch.setenv("TBK_MINIMIZER_W", str(w))
    monkeypatch.setenv("TBK_ENTRY_LOAD", "5.5" if crowded else "0.3")   # crowded: 5.5 entries per list and bucket of 8 slots
    monkeypatch.setenv("TBK_WENTRY_LOAD", "2.8" if crowded else "0.25")  # wide entries: four per list and line
    if wide:
        monkeypatch.setenv("TBK_ENTRY_WIDE", "1")
    monkeypatch.setenv("TBK_SLICE_BASES", str(int(rng.choice([2048, 5000, 1 << 30]))))
    a, b, bases, offs, want, reads = _case(rng, k, tmp_path, orc)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["entry_layout"] and st["wide_entries"] == bool(wide) and st["minimizer_w"] == w and st["sampling_t"] > 0, st
        assert st["entries_a"] + st["entries_b"] < st["distinct_a"] + st["distinct_b"], st     # runs merged
        if crowded:
            assert st["keys_behind_front"] > 0 and st["keys_past_half"] > 0, st                # second looks and walks happen
        got = cls.classify_batch(bases, offs)
        again = cls.classify_batch(bases, offs)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (k, w, wide, crowded, st, bad[:10], got[bad[:5]], want[bad[:5]], [len(reads[int(i)]) for i in bad[:5]])
    assert np.array_equal(again, want)


@pytest.mark.parametrize("k,wide", [(21, 0), (21, 1), (27, 1), (31, 1), (32, 1)])
def test_entry_layout_on_the_reference_vectors(gpu, monkeypatch, k, wide):
    """The recorded counts of the real reference (tests/golden/diff_vectors.json) through the entry layouts: k = 21 in
    narrow and in wide entries, k = 27, 31, 32 in wide ones."""
    from trio_binning_amd import kmers

    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == k)
    monkeypatch.setenv("TBK_ENTRY", "1")
    if wide:
        monkeypatch.setenv("TBK_ENTRY_WIDE", "1")
    a = kmers.HashSet.from_keys(np.array([kmers.kmer_to_int(s) for s in v["list_a"]], dtype=np.uint64), k)
    b = kmers.HashSet.from_keys(np.array([kmers.kmer_to_int(s) for s in v["list_b"]], dtype=np.uint64), k)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["entry_layout"] and st["wide_entries"] == bool(wide), st
        got = cls.classify_reads(v["reads"])
    assert np.array_equal(got, np.array(v["counts"], dtype=np.int32))


def test_clustered_lists_get_the_entry_layout(gpu, orc, monkeypatch):
    """The policy: lists shaped like find-unique-kmers output overflow the fronts of the key layout and are rebuilt as
    entries - a quarter of the slots, under 50 bytes of HBM per key; uniform lists stay in the key layout's front.  Where
    a k-mer's context does not fit a slot (k = 31) the entries are wide ones."""
    import ctypes as C

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    for v in ("TBK_ENTRY", "TBK_ENTRY_WIDE", "TBK_MOD_SAMPLING", "TBK_TABLE_LOAD", "TBK_FRONT", "TBK_MINIMIZER_W", "TBK_MINIMIZER_M", "TBK_ENTRY_LOAD", "TBK_WENTRY_LOAD", "TBK_SHORT", "TBK_SHORT_LOAD"):
        monkeypatch.delenv(v, raising=False)
    dev, n = 0, 400_000
    rng = np.random.default_rng(3)

    def hap_lists(k):
        cap = 2 * n
        ptrs = []
        for _ in range(2):
            p = C.c_void_p()
            check(lib.tbk_device_alloc(dev, cap * 8, C.byref(p)))
            ptrs.append(p.value)
        got = C.c_uint64()
        check(lib.tbk_synth_hap_keys_device(dev, 0x5EED0001, 3_000_000, int(round((1 / 500) * (1 << 24))), k, C.c_void_p(ptrs[0]), C.c_void_p(ptrs[1]), cap, C.byref(got)))
        m = got.value
        keys = np.empty(2 * m, dtype=np.uint64)
        check(lib.tbk_memcpy_d2h(dev, keys.ctypes.data, C.c_void_p(ptrs[0]), m * 8))
        check(lib.tbk_memcpy_d2h(dev, keys.ctypes.data + m * 8, C.c_void_p(ptrs[1]), m * 8))
        for p in ptrs:
            check(lib.tbk_device_free(dev, C.c_void_p(p)))
        return keys[:m], keys[m:]

    def decode(key, k):
        return "".join("ACGT"[(int(key) >> (2 * i)) & 3] for i in range(k))

    for k, want_entry in ((21, True), (23, True), (31, True)):
        ka, kb = hap_lists(k)
        oa, ob = orc.table_from_keys(ka, k), orc.table_from_keys(kb, k)
        plants = [decode(x, k) for x in np.concatenate([ka[:300], kb[:300]])]
        bases, offs = _pack(_rand_reads(rng, 300, 3000, plants, k, p_plant=0.9))
        want = orc.count_batch(bases, offs, oa, ob)
        a, b = kmers.HashSet.from_keys(ka, k), kmers.HashSet.from_keys(kb, k)
        with kmers.Classifier(a, b) as cls:
            st = cls.stats()
            assert st["entry_layout"] == want_entry and st["wide_entries"] == (k > 25), (k, st)
            if want_entry:
                keys, entries = st["distinct_a"] + st["distinct_b"], st["entries_a"] + st["entries_b"]
                assert keys > 3 * entries, (k, st)                          # a variant's windows: one entry per sampled m-mer
                assert st["table_bytes"] <= (51 if k <= 25 else 140) * (ka.size + kb.size), (k, st)
                assert st["keys_behind_front"] <= (0.08 if k <= 25 else 0.15) * entries, (k, st)   # the fronts hold them (wide entries: one per list in the front)
            else:
                assert not st["front_layout"], (k, st)
            assert np.array_equal(cls.classify_batch(bases, offs), want), k
        monkeypatch.setenv("TBK_ENTRY", "0")                                # switched off: whole lines, as before
        with kmers.Classifier(a, b) as cls:
            st = cls.stats()
            assert not st["entry_layout"] and not st["front_layout"], (k, st)
            assert np.array_equal(cls.classify_batch(bases, offs), want), k
        monkeypatch.delenv("TBK_ENTRY")
    uni = np.empty(2 * n, dtype=np.uint64)
    check(lib.tbk_synth_keys_host(0x5EED0001, 0, 2 * n, 21, uni.ctypes.data))
    # uniform lists do not merge: short keys (56 bytes of HBM per key), one build; without them the key layout's front
    ua, ub = kmers.HashSet.from_keys(uni[:n], 21), kmers.HashSet.from_keys(uni[n:], 21)
    with kmers.Classifier(ua, ub) as cls:
        st = cls.stats()
        assert st["short_keys"] and not st["entry_layout"] and not st["front_layout"] and st["layout_builds"] == 1, st
        assert st["table_bytes"] <= 58 * 2 * n and st["keys_behind_front"] <= 0.03 * 2 * n, st
    monkeypatch.setenv("TBK_SHORT", "0")
    with kmers.Classifier(ua, ub) as cls:
        st = cls.stats()
        assert not st["short_keys"] and not st["entry_layout"] and st["front_layout"] and st["layout_builds"] == 1, st


@pytest.mark.parametrize("k", [21, 25, 31])
def test_lists_from_a_genome_with_repeat_families(gpu, orc, monkeypatch, k):
    """Half of a 60-Mbase genome in 16 repeat families (~230 copies each, 2 % diverged): the copies' variant k-mers crowd
    the buckets of their family's m-mers - entries past their line, windows that walk - and reads drawn from that genome
    ask for them.  Counts equal the oracle's in the entry layout the policy picks and in the key layouts."""
    import ctypes as C

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    for v in ("TBK_ENTRY", "TBK_ENTRY_WIDE", "TBK_MOD_SAMPLING", "TBK_TABLE_LOAD", "TBK_FRONT", "TBK_MINIMIZER_W", "TBK_MINIMIZER_M", "TBK_ENTRY_LOAD", "TBK_WENTRY_LOAD"):
        monkeypatch.delenv(v, raising=False)
    dev, G, R, L = 0, 60_000_000, 160, 6000
    shape = int(round((1 / 500) * (1 << 24))) | (128 << 24)
    cap = 8_000_000

    def dalloc(n):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, n, C.byref(p)))
        return p.value

    d_keys = dalloc(2 * cap * 8)
    got = C.c_uint64()
    check(lib.tbk_synth_hap_keys_device(dev, 0x5EED0077, G, shape, k, C.c_void_p(d_keys), C.c_void_p(d_keys + cap * 8), cap, C.byref(got)))
    n = got.value
    assert 3_000_000 < n <= cap
    keys = np.empty(2 * cap, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(dev, keys.ctypes.data, C.c_void_p(d_keys), keys.nbytes))
    ka, kb = keys[:n], keys[cap:cap + n]
    d_bases, d_offs = dalloc(R * L + 32), dalloc((R + 1) * 8)
    check(lib.tbk_synth_hap_reads_device(dev, 0x5EED0077, G, shape, 0xBEEF, 0, R, L, int(0.002 * (1 << 24)), C.c_void_p(d_bases), C.c_void_p(d_offs)))
    bases, offs = np.empty(R * L, dtype=np.uint8), np.empty(R + 1, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(dev, bases.ctypes.data, C.c_void_p(d_bases), bases.nbytes))
    check(lib.tbk_memcpy_d2h(dev, offs.ctypes.data, C.c_void_p(d_offs), offs.nbytes))
    for p in (d_keys, d_bases, d_offs):
        check(lib.tbk_device_free(dev, C.c_void_p(p)))
    want = orc.count_batch(bases, offs, orc.table_from_keys(ka, k), orc.table_from_keys(kb, k))
    assert want.sum() > 20 * R                                           # the reads do find list k-mers
    a, b = kmers.HashSet.from_keys(ka, k), kmers.HashSet.from_keys(kb, k)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["entry_layout"] and st["wide_entries"] == (k > 25), st
        assert st["keys_behind_front"] > 0, st                           # crowded: entries behind their lines' fronts
        print(k, {x: st[x] for x in ("entries_a", "entries_b", "keys_behind_front", "keys_past_half", "table_bytes")})
        assert np.array_equal(cls.classify_batch(bases, offs), want)
    monkeypatch.setenv("TBK_ENTRY", "0")
    with kmers.Classifier(a, b) as cls:
        assert not cls.stats()["entry_layout"]
        assert np.array_equal(cls.classify_batch(bases, offs), want)


SHORT = [(21, 6), (21, 5), (21, 4), (19, 4), (20, 5), (22, 6), (23, 6), (23, 8), (24, 5), (24, 7), (25, 6), (25, 8)]


@pytest.mark.parametrize("k,w", SHORT)
@pytest.mark.parametrize("crowded", [0, 1])
def test_short_keys_counts_equal_the_oracle(gpu, orc, tmp_path, monkeypatch, k, w, crowded):
    """Short keys (tbk_common.h): a list k-mer as the 32 bits its bucket does not say already.  The same lists and reads as
    for the entry layouts, plus stretches of sequence with hundreds of list k-mers each (their lines overflow into the
    overflow table whatever the table's size); in the smallest table k allows (crowded: 24 keys per line asked for) and in a
    roomy one."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(1000 * k + 10 * w + crowded)
    monkeypatch.setenv("TBK_SHORT", "1")
    monkeypatch.setenv("TBK_MINIMIZER_W", str(w))
    monkeypatch.setenv("TBK_SHORT_LOAD", "24" if crowded else "0.2")
    if crowded:
        monkeypatch.setenv("TBK_SHORT_LINE_CAP", str(int(rng.choice([8, 9, 12]))))   # the inserts use that many slots of a line: the rest goes to the overflow table
    monkeypatch.setenv("TBK_SLICE_BASES", str(int(rng.choice([2048, 5000, 1 << 30]))))
    a, b, bases, offs, want, reads = _case(rng, k, tmp_path, orc, crowd_cores=6)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["short_keys"] and not st["entry_layout"] and st["minimizer_w"] == w and st["sampling_t"] > 0, st
        assert st["keys_behind_front"] > 0 and (not crowded or st["keys_past_half"] > 0), st     # second looks, and (crowded) keys in the overflow table
        got = cls.classify_batch(bases, offs)
        again = cls.classify_batch(bases, offs)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (k, w, crowded, st, bad[:10], got[bad[:5]], want[bad[:5]], [len(reads[int(i)]) for i in bad[:5]])
    assert np.array_equal(again, want)


def test_short_keys_on_the_reference_vectors(gpu, monkeypatch):
    """The recorded counts of the real reference (tests/golden/diff_vectors.json, k = 21) through short keys."""
    from trio_binning_amd import kmers

    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == 21)
    monkeypatch.setenv("TBK_SHORT", "1")
    a = kmers.HashSet.from_keys(np.array([kmers.kmer_to_int(s) for s in v["list_a"]], dtype=np.uint64), 21)
    b = kmers.HashSet.from_keys(np.array([kmers.kmer_to_int(s) for s in v["list_b"]], dtype=np.uint64), 21)
    with kmers.Classifier(a, b) as cls:
        assert cls.stats()["short_keys"]
        got = cls.classify_reads(v["reads"])
    assert np.array_equal(got, np.array(v["counts"], dtype=np.int32))


@pytest.mark.parametrize("layout", ["entry", "short"])
@pytest.mark.parametrize("k,w", [(21, 6), (21, 4), (23, 6)])
def test_sampling_over_2w_positions_still_answers(gpu, orc, tmp_path, monkeypatch, layout, k, w):
    """Narrow entries and short keys rank 3w t-mer positions per span by default (tbk_mz_span3); TBK_SPAN3=0 keeps the 2w
    the key layouts use - those kernels (LW = 2) stay in the library and answer like the oracle, crowded lines included."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(7000 + 100 * k + w + (layout == "short"))
    monkeypatch.setenv("TBK_SPAN3", "0")
    monkeypatch.setenv("TBK_MINIMIZER_W", str(w))
    if layout == "short":
        monkeypatch.setenv("TBK_SHORT", "1")
        monkeypatch.setenv("TBK_SHORT_LOAD", "24")
        monkeypatch.setenv("TBK_SHORT_LINE_CAP", "9")
    else:
        monkeypatch.setenv("TBK_ENTRY", "1")
        monkeypatch.setenv("TBK_ENTRY_LOAD", "5.5")
    a, b, bases, offs, want, reads = _case(rng, k, tmp_path, orc, crowd_cores=3)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["short_keys"] == (layout == "short") and st["entry_layout"] == (layout == "entry"), st
        assert st["sampling_t"] == st["minimizer_m"] - st["minimizer_w"], st          # 2w positions: t = m - w
        assert st["keys_behind_front"] > 0 and st["keys_past_half"] > 0, st
        got = cls.classify_batch(bases, offs)
    assert np.array_equal(got, want)
    monkeypatch.setenv("TBK_SPAN3", "1")
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        assert st["sampling_t"] == st["minimizer_m"] - 2 * st["minimizer_w"], st      # 3w positions: t = m - 2w
        assert np.array_equal(cls.classify_batch(bases, offs), want)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TBK_FUZZ_SEEDS", "24"))))  # more seeds for a soak run
def test_seeded_fuzz_of_the_entry_and_short_key_layouts(gpu, orc, tmp_path, seed, monkeypatch):
    """Random small configurations of the round-4 layouts: any k they hold, any span, 2w or 3w t-mer positions, tables from
    roomy to crowded, lists full of what makes ranks tie and m-mers palindromic (runs of one base, short-period repeats,
    reverse complements, duplicates, lines shared between the lists), reads with N and lower case and ragged lengths."""
    from trio_binning_amd import kmers

    rng = np.random.default_rng(31000 + seed)
    layout = ["entry", "short", "wide"][seed % 3]
    k = int(rng.choice({"entry": [17, 19, 21, 22, 23, 24, 25], "short": [17, 19, 20, 21, 22, 23, 24, 25], "wide": [21, 26, 27, 29, 31, 32]}[layout]))
    monkeypatch.setenv("TBK_MINIMIZER_W", str(int(rng.integers(2, 9))))
    monkeypatch.setenv("TBK_SPAN3", str(int(rng.integers(0, 2))))
    monkeypatch.setenv("TBK_SLICE_BASES", str(int(rng.choice([2048, 5000, 1 << 30]))))
    if layout == "short":
        monkeypatch.setenv("TBK_SHORT", "1")
        monkeypatch.setenv("TBK_SHORT_LOAD", str(rng.choice([0.1, 2.3, 24])))
        monkeypatch.setenv("TBK_SHORT_LINE_CAP", str(int(rng.choice([2, 8, 9, 32]))))
    else:
        monkeypatch.setenv("TBK_ENTRY", "1")
        monkeypatch.setenv("TBK_ENTRY_LOAD", str(rng.choice([0.2, 0.5, 5.5])))
        monkeypatch.setenv("TBK_WENTRY_LOAD", str(rng.choice([0.1, 0.25, 2.8])))
        if layout == "wide":
            monkeypatch.setenv("TBK_ENTRY_WIDE", "1")
    n_a, n_b = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))

    def rand_kmer():
        mode = rng.random()
        if mode < 0.15:   # low complexity: long runs of one base
            return ("ACGT"[int(rng.integers(0, 4))] * k)[: int(rng.integers(0, k + 1))].ljust(k, "ACGT"[int(rng.integers(0, 4))])
        if mode < 0.3:    # short-period repeat
            unit = "".join("ACGT"[c] for c in rng.integers(0, 4, int(rng.integers(1, 5))))
            return (unit * k)[:k]
        if mode < 0.4:    # its own reverse complement in the middle
            half = "".join("ACGT"[c] for c in rng.integers(0, 4, (k + 1) // 2))
            return (half + _rc(half))[:k]
        return "".join("ACGT"[c] for c in rng.integers(0, 4, k))

    la = [rand_kmer() for _ in range(n_a)]
    lb = [rand_kmer() for _ in range(n_b)]
    run = "".join("ACGT"[c] for c in rng.integers(0, 4, 400))                 # runs of overlapping k-mers, as find-unique-kmers writes them
    la += [run[i:i + k] for i in range(0, 150)]
    lb += [_rc(run[i:i + k]) for i in range(200, 350)]
    lb += [la[int(i)] for i in rng.integers(0, n_a, min(20, n_a))]          # shared with hapA
    lb += [_rc(la[int(i)]) for i in rng.integers(0, n_a, min(10, n_a))]     # shared, other strand
    la += [la[int(i)] for i in rng.integers(0, n_a, 5)]                     # duplicates
    fa = _write(tmp_path, "a.txt", "".join(x + "\n" for x in la))
    fb = _write(tmp_path, "b.txt", "\n".join(lb))
    oa, ob = orc.table_from_file(fa), orc.table_from_file(fb)
    a, b = kmers.HashSet.from_file(fa), kmers.HashSet.from_file(fb)
    reads = _rand_reads(rng, int(rng.integers(1, 100)), int(rng.choice([40, 300, 2500, 9000])), la + lb, k, p_plant=0.9)
    reads += ["", "A" * max(0, k - 1), la[0], _rc(lb[0]) * 2, "".join(la[:40]), "".join(_rc(x) for x in lb[:40]), run, _rc(run), "A" * 500, "AT" * 300, "ACG" * 200]
    body = "".join("ACGT"[c] for c in rng.integers(0, 4, 5000))
    reads += [body[: 2048 - sum(map(len, reads)) % 2048], body[:2047], body[:2048 + k - 1], body[:4096]]  # pass-boundary shapes
    noisy = list(body[:3000])
    for i in rng.integers(0, 3000, 40):
        noisy[int(i)] = "NnacgtR-"[int(rng.integers(0, 8))]
    reads.append("".join(noisy))
    reads = [reads[int(i)] for i in rng.permutation(len(reads))]
    bases, offs = _pack(reads)
    want = orc.count_batch(bases, offs, oa, ob, strict=True)
    with kmers.Classifier(a, b) as cls:
        st = cls.stats()
        # (a span the layout asked for has no room for at this k - the span is drawn at random - ends in wide entries or in the key
        # layouts: which layout stands is not asserted here, the parametrized tests above do that; what is, is the counts)
        got = cls.classify_batch(bases, offs)
        again = cls.classify_batch(bases, offs)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (seed, layout, k, st, bad[:10], got[bad[:5]], want[bad[:5]])
    assert np.array_equal(again, want)
