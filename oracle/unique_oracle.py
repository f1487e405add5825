"""CPU restatement of the find-unique-kmers step — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Two layers with different standing:

* ``analyze_histogram_rows`` follows the reference's own arithmetic (find_unique_kmers.py:132-168:
  first local minimum of the histogram, then the first count whose row drops below it) and is
  PINNED: tests/golden/unique_cutoffs.json was recorded from the real function.

* ``count_kmers`` / ``histogram_rows`` / ``unique_kmers`` restate what the KMC 3 tools the reference
  shells out to do at its settings (find_unique_kmers.py:82-90,123-129,186-194,218-225): canonical
  counting over both strands, k-mers with a symbol outside ACGT skipped, lower case = upper case,
  default -ci2 (k-mers seen once are not stored), -cs255 (counters saturate), kmers_subtract, dump
  with -ci/-cx in lexicographic order.  KMC is not in the reference checkout and not installed:
  PARITY WITH KMC IS UNPINNED; these functions pin the GPU path to this stated reading of it.

Pure Python: small inputs only.
"""
from collections import Counter
from typing import Dict, Iterable, List, Optional, Tuple

_COMP = str.maketrans("ACGT", "TGCA")


def canonical(kmer: str) -> str:
    rc = kmer.translate(_COMP)[::-1]
    return kmer if kmer <= rc else rc


def count_kmers(reads: Iterable[str], k: int) -> Counter:
    """Occurrences of every canonical k-mer (uncapped, singletons included)."""
    c: Counter = Counter()
    for read in reads:
        s = read.upper()
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if all(ch in "ACGT" for ch in w):
                c[canonical(w)] += 1
    return c


def database(counts: Counter) -> Dict[str, int]:
    """What `kmc` leaves in its database: counter >= 2 (-ci2), saturated at 255 (-cs255)."""
    return {km: min(n, 255) for km, n in counts.items() if n >= 2}


def histogram_rows(db: Dict[str, int]) -> List[Tuple[int, int]]:
    """Rows of `kmc_tools transform db histogram`: (counter value, number of k-mers), values 1..255."""
    h = Counter(db.values())
    return [(c, h.get(c, 0)) for c in range(1, 256)]


class HistogramError(Exception):
    pass


def analyze_histogram_rows(rows: Iterable[Tuple[int, int]]) -> Tuple[int, int, bool]:
    """(min_coverage, max_coverage, warned) from histogram rows — find_unique_kmers.py:132-168."""
    min_cov, max_cov = False, False
    min_cov_count = None
    last = -1
    for coverage, count in rows:
        if coverage != 2:  # the row of count 2 is only remembered (:136)
            if not min_cov:
                if count > last:  # counts start rising: the row before is the local minimum (:140-143)
                    min_cov = coverage - 1
                    min_cov_count = last
            elif not max_cov:
                if count < min_cov_count:  # first row below the count at the minimum (:147-150)
                    max_cov = coverage
                    break
        last = count
    if not min_cov or not max_cov:  # 0 counts as "not found", as in the reference (:154)
        raise HistogramError()
    return min_cov, max_cov, (max_cov - min_cov < 5)


def unique_kmers(db_a: Dict[str, int], db_b: Dict[str, int], min_count: int, max_count: int) -> List[str]:
    """kmers_subtract then kmc_dump -ci -cx: k-mers of A not in B with min <= counter <= max, sorted."""
    return sorted(km for km, n in db_a.items() if km not in db_b and min_count <= n <= max_count)
