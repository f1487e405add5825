"""find-unique-kmers step on the GPU against the oracle's restatement of the KMC steps (parity with
KMC itself is unpinned: see oracle/unique_oracle.py), through the C-ABI and through the CLI."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = str.maketrans("ACGT", "TGCA")


def _rc(s):
    return s.translate(COMP)[::-1]


def _library(rng, genome, n_reads, read_len, err=0.01, lower=0.0, n_rate=0.001):
    reads = []
    for _ in range(n_reads):
        p = int(rng.integers(0, len(genome) - read_len))
        s = list(genome[p:p + read_len])
        for i in np.nonzero(rng.random(read_len) < err)[0]:
            s[int(i)] = "ACGT"[int(rng.integers(0, 4))]
        for i in np.nonzero(rng.random(read_len) < n_rate)[0]:
            s[int(i)] = "N"
        r = "".join(s)
        if rng.random() < 0.5:
            r = _rc(r)
        if rng.random() < lower:
            r = r.lower()
        reads.append(r)
    return reads


def _two_parents(rng, glen=30_000, snp=1 / 200):
    base = "".join("ACGT"[c] for c in rng.integers(0, 4, glen))
    def mutate():
        s = list(base)
        for i in np.nonzero(rng.random(glen) < snp)[0]:
            s[int(i)] = "ACGT"[(("ACGT".index(s[int(i)])) + int(rng.integers(1, 4))) % 4]
        return "".join(s)
    return mutate(), mutate()


@pytest.mark.parametrize("k", [5, 16, 21, 31, 32])
def test_counter_matches_oracle(gpu, tmp_path, k):
    from oracle import unique_oracle as uo
    from trio_binning_amd import kmers

    rng = np.random.default_rng(100 + k)
    ga, gb = _two_parents(rng, glen=8_000 if k > 5 else 600)
    reads_a = _library(rng, ga, 900, 150, lower=0.1) + ["", "ACGT", "N" * 40, ga[:k], ga[:k]]
    reads_b = _library(rng, gb, 700, 150)
    with kmers.KmerCounter(k, 400_000) as ca, kmers.KmerCounter(k, 400_000) as cb:
        for i in range(0, len(reads_a), 250):  # several batches accumulate
            ca.add_reads(reads_a[i:i + 250])
        cb.add_reads(reads_b)
        oa, ob = uo.count_kmers(reads_a, k), uo.count_kmers(reads_b, k)
        dba, dbb = uo.database(oa), uo.database(ob)
        hist = ca.histogram()
        assert int(hist[0]) == len(oa)
        want_rows = uo.histogram_rows(dba)
        assert [int(hist[c]) for c in range(2, 256)] == [n for c, n in want_rows if c >= 2]
        assert int(hist[1]) == sum(1 for n in oa.values() if n == 1)
        for lo, hi in ((2, 255), (3, 20), (5, 5), (1, 4), (200, 255)):
            out = str(tmp_path / f"u_{lo}_{hi}.txt")
            n = ca.unique(cb, lo, hi, out)
            got = open(out).read().split("\n")
            assert got[-1] == "" and got[:-1] == uo.unique_kmers(dba, dbb, lo, hi) and n == len(got) - 1
        st = ca.stats()
        assert st["reads_added"] == len(reads_a) and st["bases_added"] == sum(map(len, reads_a))


@pytest.mark.parametrize("seed", range(int(os.environ.get("TBK_FUZZ_SEEDS", "8"))))
def test_counter_seeded_fuzz(gpu, tmp_path, seed):
    """Random small libraries: any k, read lengths from 0 to a few hundred, N and lower case, batches
    of random size, counters starting too small or roomy; histogram, distinct count and three dumps
    against the oracle."""
    from oracle import unique_oracle as uo
    from trio_binning_amd import kmers

    rng = np.random.default_rng(5000 + seed)
    k = int(rng.choice([1, 3, 8, 15, 16, 17, 21, 25, 31, 32]))
    ga, gb = _two_parents(rng, glen=int(rng.choice([300, 3000, 12000])), snp=1 / 100)
    def lib(g):
        n, L = int(rng.integers(1, 400)), int(rng.choice([20, 75, 150, 400]))
        L = min(L, len(g) - 1)
        reads = _library(rng, g, n, L, err=float(rng.choice([0.0, 0.01, 0.05])), lower=0.2, n_rate=0.003)
        return reads + ["", "N" * 30, g[:max(k - 1, 0)], g[:k], g[:k].lower()]
    reads_a, reads_b = lib(ga), lib(gb)
    cap_a, cap_b = int(rng.choice([16, 1000, 200_000])), int(rng.choice([16, 200_000]))
    with kmers.KmerCounter(k, cap_a) as ca, kmers.KmerCounter(k, cap_b) as cb:
        i = 0
        while i < len(reads_a):
            step = int(rng.integers(1, 200))
            ca.add_reads(reads_a[i:i + step])
            i += step
        cb.add_reads(reads_b)
        oa, ob = uo.count_kmers(reads_a, k), uo.count_kmers(reads_b, k)
        dba, dbb = uo.database(oa), uo.database(ob)
        hist = ca.histogram()
        assert int(hist[0]) == len(oa) == ca.stats()["distinct"]
        assert [int(hist[c]) for c in range(2, 256)] == [n for c, n in uo.histogram_rows(dba) if c >= 2]
        for lo, hi in ((2, 255), (3, 9), (int(rng.integers(1, 6)), int(rng.integers(6, 300)))):
            out = str(tmp_path / f"u_{lo}_{hi}.txt")
            n = ca.unique(cb, lo, hi, out)
            got = open(out).read().split("\n")
            assert got[:-1] == uo.unique_kmers(dba, dbb, lo, min(hi, 255)) and n == len(got) - 1, (k, lo, hi)


def test_counter_grows_before_a_batch_could_fill_it(gpu, tmp_path):
    """A counter created far too small is rebuilt larger (several times) as batches arrive; the
    counters it already holds move with it."""
    from oracle import unique_oracle as uo
    from trio_binning_amd import kmers

    k = 21
    rng = np.random.default_rng(1)
    genome = "".join("ACGT"[c] for c in rng.integers(0, 4, 20_000))
    reads = _library(rng, genome, 3000, 120, err=0.02)
    with kmers.KmerCounter(k, 1000) as c, kmers.KmerCounter(k, 1000) as empty:
        slots0 = c.stats()["n_slots"]
        for i in range(0, len(reads), 500):
            c.add_reads(reads[i:i + 500])
        st = c.stats()
        want = uo.count_kmers(reads, k)
        assert st["n_slots"] > 50 * slots0 and st["distinct"] == len(want)
        hist = c.histogram()
        assert int(hist[0]) == len(want)
        out = str(tmp_path / "all.txt")
        n = c.unique(empty, 2, 255, out)
        assert open(out).read().split() == uo.unique_kmers(uo.database(want), {}, 2, 255) and n > 1000


def test_find_unique_kmers_cli(gpu, tmp_path, capsys):
    """Two parents of one 60 kb genome at ~25x: the CLI's lists equal the oracle's, the histograms
    have kmc_tools' shape, and the lists bin reads of either haplotype correctly."""
    from oracle import unique_oracle as uo
    from trio_binning_amd import find_unique_kmers as fu
    from trio_binning_amd import kmers

    k = 21
    rng = np.random.default_rng(77)
    ga, gb = _two_parents(rng, glen=60_000)
    reads_a, reads_b = _library(rng, ga, 10_000, 150), _library(rng, gb, 10_000, 150)

    def fastq(path, reads, gz=False):
        text = "".join(f"@r{i} x\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads))
        (gzip.open if gz else open)(path, "wt").write(text)
        return str(path)

    fa1, fa2 = fastq(tmp_path / "a1.fastq", reads_a[:6000]), fastq(tmp_path / "a2.fastq.gz", reads_a[6000:], gz=True)
    fb = fastq(tmp_path / "b.fastq", reads_b)
    out = tmp_path / "out"
    out.mkdir()
    fu.main(["-k", str(k), "-o", str(out), "-s", str(tmp_path), "--capacity", "3000000", fa1 + "," + fa2, fb])
    err = capsys.readouterr().err
    assert "Using counts in range [" in err and "# of unique k-mers in haplotype A:" in err

    dba, dbb = uo.database(uo.count_kmers(reads_a, k)), uo.database(uo.count_kmers(reads_b, k))
    for name, db, other in (("A", dba, dbb), ("B", dbb, dba)):
        rows = [tuple(map(int, l.split("\t"))) for l in open(tmp_path / f"haplotype{name}.histogram")]
        assert rows == uo.histogram_rows(db)
        lo, hi, _ = uo.analyze_histogram_rows(rows)
        got = open(out / f"hap{name}_only_kmers.txt").read().split()
        assert got == uo.unique_kmers(db, other, lo, hi) and len(got) > 1000
    # the lists do their job: reads drawn from each parent's genome land in that parent's bin
    a, b = kmers.HashSet.from_file(str(out / "hapA_only_kmers.txt")), kmers.HashSet.from_file(str(out / "hapB_only_kmers.txt"))
    long_a = [ga[p:p + 5000] for p in range(0, 50_000, 5000)]
    long_b = [gb[p:p + 5000] for p in range(0, 50_000, 5000)]
    with kmers.Classifier(a, b) as cls:
        counts = cls.classify_batch(*kmers.pack_reads(long_a + long_b))
    assert (counts[:10, 0] > counts[:10, 1]).all() and (counts[10:, 1] > counts[10:, 0]).all()
