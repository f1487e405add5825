"""BASELINE.json's table sizes on one GPU, at full scale: configs[1] (2 x 100 M 21-mers), configs[2]
(2 x 300 M 21-mers, the headline), configs[4] (2 x 1 B 31-mers with 100 kb reads: the 64-bit m-mer
kernel, m = 18) on BASELINE's uniform lists, and configs[4]'s size again on lists shaped like real
find-unique-kmers output (runs of overlapping k-mers around SNPs: dense bucket overflow and walks).

Each configuration is checked two ways:

* against the oracle, count for count, on a sample of the reads (the oracle's tables are built from
  the very keys the GPU generated, copied back; only the sample is bounded - the oracle does about
  1.5 Mbases/s per core);
* through size-independent properties on all reads:
    strand symmetry  a read and its reverse complement have identical (hapA, hapB) counts, because
                     every window is looked up by its canonical k-mer;
    splitting        cutting every read into two pieces that overlap by k-1 bases preserves the sum;
    permutation      shuffling the read order permutes the counts and nothing else;
    determinism      two launches give identical counts;
    generator promise  origin reads carry at least their 30 planted k-mers / haplotype reads side
                     with their own haplotype."""
import ctypes as C
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY_SEED, READ_SEED = 0x5EED0001, 0x5EED0002
K, N_LIST = 21, 300_000_000   # the headline configuration (used by the counter test below as well)

CONFIGS = {
    "configs1_2x100M_k21": dict(k=21, n=100_000_000, R=8192, L=15_000, lists="uniform", m_gt_16=False, sample=256),
    "configs2_2x300M_k21": dict(k=21, n=300_000_000, R=16_384, L=15_000, lists="uniform", m_gt_16=False, sample=256),
    "configs4_2x1B_k31": dict(k=31, n=1_000_000_000, R=2_048, L=100_000, lists="uniform", m_gt_16=True, sample=96),
    "configs4_haplotype_lists_k31": dict(k=31, n=1_000_000_000, R=1_024, L=100_000, lists="haplotypes", m_gt_16=True, sample=64),
}
SNP_RATE, ERR_RATE = 1 / 500, 0.002


@pytest.fixture(scope="module", params=list(CONFIGS))
def big(request, gpu, orc):
    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    cfg = dict(CONFIGS[request.param], name=request.param)
    dev, k, n, R, L = 0, cfg["k"], cfg["n"], cfg["R"], cfg["L"]

    def dalloc(nbytes):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, nbytes, C.byref(p)))
        return p.value

    hap = cfg["lists"] == "haplotypes"
    if hap:
        snp24, err24 = int(round(SNP_RATE * (1 << 24))), int(round(ERR_RATE * (1 << 24)))
        p_diff = 2 * SNP_RATE - SNP_RATE ** 2 * (1 + 1 / 3)
        genome_len = int(n / (1 - (1 - p_diff) ** k))
        stride = int(n * 1.05) + 1024
        d_keys = dalloc(2 * stride * 8)
        got = C.c_uint64()
        check(lib.tbk_synth_hap_keys_device(dev, KEY_SEED, genome_len, snp24, k, C.c_void_p(d_keys), C.c_void_p(d_keys + stride * 8),
                                            stride, C.byref(got)))
        assert 0.9 * n < got.value <= stride
        n = got.value
    else:
        stride = n
        d_keys = dalloc(2 * n * 8)
        check(lib.tbk_synth_keys_device(dev, KEY_SEED, 0, 2 * n, k, C.c_void_p(d_keys)))
    a = kmers.HashSet.from_device_keys(d_keys, n, k)
    b = kmers.HashSet.from_device_keys(d_keys + stride * 8, n, k)
    cls = kmers.Classifier(a, b)
    a.close()
    b.close()  # the classifier owns the hashed tables; the lists may go
    st = cls.stats()
    if not hap:
        assert st["distinct_a"] == n and st["distinct_b"] == n
    assert (st["minimizer_m"] > 16) == cfg["m_gt_16"], st      # configs[4] runs the 64-bit m-mer kernel
    # the oracle's tables from the same keys (host copies are dropped as soon as the tables stand)
    threads = kmers.host_threads()
    tables = []
    for which in (0, 1):
        h = np.empty(n, dtype=np.uint64)
        check(lib.tbk_memcpy_d2h(dev, h.ctypes.data, C.c_void_p(d_keys + which * stride * 8), n * 8))
        tables.append(orc.table_from_keys(h, k, threads=threads))
        del h
    check(lib.tbk_device_free(dev, C.c_void_p(d_keys)))
    total = R * L
    d_bases, d_offs = dalloc(total + 64), dalloc((R + 1) * 8)
    if hap:
        check(lib.tbk_synth_hap_reads_device(dev, KEY_SEED, genome_len, snp24, READ_SEED, 0, R, L, err24, C.c_void_p(d_bases), C.c_void_p(d_offs)))
    else:
        check(lib.tbk_synth_reads_device(dev, READ_SEED, 0, R, L, KEY_SEED, n, n, k, 30, 3, C.c_void_p(d_bases), C.c_void_p(d_offs)))
    bases = np.empty(total, dtype=np.uint8)
    check(lib.tbk_memcpy_d2h(dev, bases.ctypes.data, C.c_void_p(d_bases), total))
    for p in (d_bases, d_offs):
        check(lib.tbk_device_free(dev, C.c_void_p(p)))
    offs = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    base_counts = cls.classify_batch(bases, offs)
    cfg.update(stats=st, n=n, threads=threads)
    yield cls, bases, offs, base_counts, cfg, tables
    cls.close()
    del tables, bases
    gc.collect()


def test_sample_equals_the_oracle(big, orc):
    """c/kmers.c:270-299 restated (oracle/kmers_oracle.c), same tables, a sample of whole reads from
    both ends of the batch."""
    cls, bases, offs, counts, cfg, (oa, ob) = big
    R, L, s = cfg["R"], cfg["L"], cfg["sample"]
    head = orc.count_batch(bases[: s // 2 * L], offs[: s // 2 + 1], oa, ob, threads=cfg["threads"])
    tail = orc.count_batch(bases[(R - s // 2) * L:], offs[: s // 2 + 1], oa, ob, threads=cfg["threads"])
    assert np.array_equal(counts[: s // 2], head), cfg["name"]
    assert np.array_equal(counts[R - s // 2:], tail), cfg["name"]
    assert head.sum() + tail.sum() > 30 * s // 2


def test_generator_promise(big):
    cls, bases, offs, counts, cfg, _ = big
    R = cfg["R"]
    if cfg["lists"] == "uniform":
        major = counts.max(axis=1)
        assert (major >= 30).sum() > 0.85 * R          # origin A or B: p = 0.9
        # origin-less reads: 3 + 3 planted plus a few chance hits per list (k = 21: 6e8 keys in a 2.2e12 space)
        assert ((major >= 30) | (major <= 20)).all()
        assert counts.sum() > 30 * 0.85 * R
    else:
        # read r is drawn from haplotype r & 1 (0 = A): its own list must win by a wide margin
        own = np.where(np.arange(R) % 2 == 0, counts[:, 0], counts[:, 1])
        other = np.where(np.arange(R) % 2 == 0, counts[:, 1], counts[:, 0])
        assert (own > 4 * other + 50).mean() > 0.99


def test_deterministic(big):
    cls, bases, offs, counts, cfg, _ = big
    assert np.array_equal(cls.classify_batch(bases, offs), counts)


def test_strand_symmetry(big):
    cls, bases, offs, counts, cfg, _ = big
    R, L = cfg["R"], cfg["L"]
    comp = np.zeros(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    rc = comp[bases.reshape(R, L)[:, ::-1]].reshape(-1)
    assert np.array_equal(cls.classify_batch(np.ascontiguousarray(rc), offs), counts)


def test_split_reads_preserve_the_sum(big):
    cls, bases, offs, counts, cfg, _ = big
    R, L, k = cfg["R"], cfg["L"], cfg["k"]
    h = L // 2
    m = bases.reshape(R, L)
    left, right = m[:, : h + k - 1], m[:, h:]          # windows 0..h-1 and h..L-k
    pieces = np.concatenate([np.ascontiguousarray(left).reshape(-1), np.ascontiguousarray(right).reshape(-1)])
    lens = np.concatenate([np.full(R, h + k - 1, dtype=np.uint64), np.full(R, L - h, dtype=np.uint64)])
    poffs = np.zeros(2 * R + 1, dtype=np.uint64)
    np.cumsum(lens, out=poffs[1:])
    got = cls.classify_batch(pieces, poffs)
    assert np.array_equal(got[:R] + got[R:], counts)


def test_read_order_permutation(big):
    cls, bases, offs, counts, cfg, _ = big
    R, L = cfg["R"], cfg["L"]
    perm = np.random.default_rng(1).permutation(R)
    shuffled = np.ascontiguousarray(bases.reshape(R, L)[perm]).reshape(-1)
    assert np.array_equal(cls.classify_batch(shuffled, offs), counts[perm])


def test_counter_properties_at_scale(gpu):
    """The k-mer counter of the find-unique-kmers step on 0.5 Gbases of synthetic short reads
    (1e8 distinct 21-mers), checked through properties that need no oracle: the histogram adds up
    to the distinct count; counting the same reads again doubles every counter; the reverse
    complements of the reads count the same canonical k-mers; a library is 'unique' against an
    empty one exactly where its own counter window says so, and never against itself."""
    import os
    import tempfile

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    dev, k, L = 0, 21, 150
    Rn = 3_300_000

    def dalloc(n):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, n, C.byref(p)))
        return p.value

    d_bases, d_offs = dalloc(Rn * L + 64), dalloc((Rn + 1) * 8)
    check(lib.tbk_synth_hap_reads_device(dev, KEY_SEED, 50_000_000, 0, READ_SEED, 0, Rn, L, int(0.002 * (1 << 24)),
                                         C.c_void_p(d_bases), C.c_void_p(d_offs)))
    with kmers.KmerCounter(k, 200_000_000) as once, kmers.KmerCounter(k, 1 << 16) as twice, kmers.KmerCounter(k, 1 << 20) as rc, \
            kmers.KmerCounter(k, 1 << 16) as empty:
        once.add_device(d_bases, d_offs, Rn, Rn * L)
        h1 = once.histogram().astype(np.int64)
        assert h1[0] == h1[1:].sum() == once.stats()["distinct"] and h1[0] > 50_000_000
        assert h1[1] > 1_000_000 and int(np.argmax(h1[3:60])) + 3 in range(7, 13)  # error k-mers; coverage peak near 10 x 130/150
        for _ in range(2):  # a table that starts tiny and grows many times
            twice.add_device(d_bases, d_offs, Rn, Rn * L)
        h2 = twice.histogram().astype(np.int64)
        assert h2[0] == h1[0] and h2[1] == 0
        assert np.array_equal(h2[2:255:2], h1[1:128]) and h2[3:255:2].sum() == 0 and h2[255] == h1[128:].sum()
        # reverse complements: same canonical k-mers
        host = np.empty(Rn * L, dtype=np.uint8)
        check(lib.tbk_memcpy_d2h(dev, host.ctypes.data, C.c_void_p(d_bases), host.size))
        comp = np.zeros(256, dtype=np.uint8)
        comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
        rev = np.ascontiguousarray(comp[host.reshape(Rn, L)[:, ::-1]].reshape(-1))
        offs = np.arange(Rn + 1, dtype=np.uint64) * np.uint64(L)
        rc.add(rev, offs)
        assert np.array_equal(rc.histogram().astype(np.int64), h1)
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "u.txt")
            assert once.unique(empty, 5, 20, out) == h1[5:21].sum() and os.path.getsize(out) == h1[5:21].sum() * (k + 1)
            assert once.unique(once, 2, 255, out) == 0 and os.path.getsize(out) == 0
            assert once.unique(rc, 2, 255, out) == 0
    for p in (d_bases, d_offs):
        check(lib.tbk_device_free(dev, C.c_void_p(p)))
