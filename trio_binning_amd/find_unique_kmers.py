"""find-unique-kmers -- given two short-read libraries, find k-mers that are unique to each.

Host driver of the MI355X path for the reference's find_unique_kmers.py: same command line, same
output files (`hapA_only_kmers.txt`, `hapB_only_kmers.txt` under --outpath), same choice of count
cut-offs; the KMC subprocesses (`kmc`, `kmc_tools transform ... histogram`, `kmc_tools simple ...
kmers_subtract`, `kmc_dump`; find_unique_kmers.py:62-233) are replaced by a counting table in HBM
behind the C-ABI (`tbk_counter_*`).  What KMC does at those call sites is restated from its
documentation (canonical counting, -ci2, -cs255, lexicographic dump); KMC is not part of the
reference checkout, so equality with its output is not pinned by any fixture.
"""
import argparse
import os
import sys
from typing import List, Sequence, Tuple

from . import _lib

_lib.warm_up()  # the HIP runtime starts beside the imports and the argument parsing below

from . import kmers, seq  # noqa: E402

_BATCH_BASES = int(os.environ.get("TBK_BATCH_BASES", str(256 << 20)))
_BATCH_READS = int(os.environ.get("TBK_BATCH_READS", str(4 << 20)))


class HistogramError(Exception):
    """Same message as the reference's (find_unique_kmers.py:15-22), minus the command to re-run."""

    def __init__(self, histogram_path: str):
        self.message = (
            "Could not find min and max counts in histogram. "
            + "Take a look at the histogram in {} and choose cutoffs manually.".format(histogram_path)
        )
        super().__init__(self.message)


def parse_args(argv=None):
    """Same options as the reference (find_unique_kmers.py:25-59); --path-to-kmc and --threads are
    accepted and ignored (there is no kmc to find or to give threads to)."""
    parser = argparse.ArgumentParser(
        description="Given multiple short-read libraries, find k-mers that are unique to each library."
    )
    parser.add_argument("-k", "--kmer-size", type=int, required=True)
    parser.add_argument("--path-to-kmc", default="kmc", help="ignored: k-mers are counted on the GPU")
    parser.add_argument("-p", "--threads", type=int, default=1, help="ignored: k-mers are counted on the GPU")
    parser.add_argument("-o", "--outpath", default=".", help="Prefix to write output haplotypes to")
    parser.add_argument("-s", "--scratch-dir", default=".", help="Directory for the count histograms")
    parser.add_argument(
        "--capacity", type=int, default=0,
        help="distinct k-mers each counting table must hold (sequencing errors included); default: an estimate "
             "from the input sizes, within the free HBM",
    )
    parser.add_argument(
        "read_files", nargs=2,
        help="one comma-separated list of file paths for both libraries being compared. Files can "
             "be in fasta or fastq format, and uncompressed or gzipped.",
    )
    return parser.parse_args(argv)


def analyze_histogram(rows: Sequence[Tuple[int, int]], histogram_path: str = "") -> Tuple[int, int]:
    """Choose the minimum and maximum k-mer count from histogram rows (count, number of k-mers), as
    the reference does (find_unique_kmers.py:132-168): the minimum is the row before the counts first
    rise again (the row of count 2 is only remembered), the maximum the first later row that drops
    below the count at the minimum.  Raises HistogramError when either is not found (0 counts as not
    found, as in the reference); warns on stderr when they are less than 5 apart."""
    low = high = 0        # 0 doubles as "not found yet", which is how the reference treats a cut-off of 0
    floor = None          # number of k-mers in the row of the minimum
    previous = -1         # number of k-mers in the row before this one
    for count, n_kmers in rows:
        if count == 2:    # the row of count 2 is only remembered
            previous = n_kmers
            continue
        if not low:
            if n_kmers > previous:       # the histogram starts rising: the row before is the minimum
                low, floor = count - 1, previous
        elif n_kmers < floor:            # first row after the peak that falls below the minimum's row
            high = count
            break
        previous = n_kmers
    if not low or not high:
        raise HistogramError(histogram_path)
    min_coverage, max_coverage = low, high
    if max_coverage - min_coverage < 5:
        print(
            "WARNING: min and max coverage not very far apart. This may be a result of coverage being too low. "
            'Try taking a look at the histogram in "{}" yourself.'.format(histogram_path),
            file=sys.stderr,
        )
    return min_coverage, max_coverage


def count_library(paths: List[str], k: int, capacity: int) -> "kmers.KmerCounter":
    """Count the canonical k-mers of all files of one library (what `kmc -k<k> @files` does).
    The files are read side by side, one reader thread each (a gzip stream inflates on one core, but
    a library usually comes as many files); this thread feeds their batches to the GPU."""
    import queue
    import threading

    counter = kmers.KmerCounter(k, capacity)
    n_readers = max(1, min(len(paths), kmers.host_threads()))
    todo: "queue.Queue" = queue.Queue()
    for p in paths:
        todo.put(p)
    filled: "queue.Queue" = queue.Queue(maxsize=2 * n_readers)
    failure: List[BaseException] = []
    batches: List[seq.Batch] = []  # every batch made; closed by this thread at the end

    def read_files() -> None:
        free: "queue.Queue" = queue.Queue()
        for _ in range(2):
            b = seq.Batch()
            batches.append(b)
            free.put(b)
        try:
            while not failure:
                try:
                    path = todo.get_nowait()
                except queue.Empty:
                    break
                reader = seq.BatchReader(path)
                try:
                    while not failure:
                        batch = free.get()
                        if not reader.next_batch(batch, _BATCH_BASES, _BATCH_READS):
                            free.put(batch)
                            break
                        filled.put((batch, free))
                finally:
                    reader.close()
        except BaseException as exc:  # handed to the counting thread
            failure.append(exc)
        finally:
            filled.put(None)

    threads = [threading.Thread(target=read_files, name="tbk-reader-%d" % i, daemon=True) for i in range(n_readers)]
    for t in threads:
        t.start()
    live = n_readers
    try:
        while live:
            item = filled.get()
            if item is None:
                live -= 1
                continue
            batch, free = item
            if not failure:
                try:
                    counter.add_batch(batch)
                except BaseException as exc:
                    failure.append(exc)
            free.put(batch)
    finally:
        for t in threads:
            t.join(timeout=5)
        for b in batches:
            b.close()
    if failure:
        counter.close()
        raise failure[0]
    return counter


def estimate_capacity(paths: List[str]) -> int:
    """Distinct k-mers cannot outnumber the bases: uncompressed FASTQ spends two bytes per base,
    gzip compresses it about fourfold.  Bounded by what two tables (16 bytes per slot at load 0.6)
    may take of the free HBM."""
    bases = 0
    for p in paths:
        size = os.path.getsize(p)
        bases += size * 2 if p.endswith(".gz") else size // 2 + 1
    free, _total = kmers.device_mem_info()
    fit = int(0.35 * free / 16 * 0.6)
    return max(1 << 16, min(bases, fit))


def write_histogram(path: str, hist: Sequence[int]) -> List[Tuple[int, int]]:
    """The rows `kmc_tools transform <db> histogram` writes: count <tab> number of k-mers; the
    database holds no k-mer seen once (kmc's default -ci2)."""
    rows = [(c, 0 if c == 1 else int(hist[c])) for c in range(1, 256)]
    with open(path, "w") as fh:
        for c, n in rows:
            fh.write("{}\t{}\n".format(c, n))
    return rows


def main(argv=None):
    args = parse_args(argv)
    k = args.kmer_size
    libraries = []  # (counter, min_count, max_count)
    try:
        for hap_id, files_string in zip(["A", "B"], args.read_files):
            print("\033[92mCounting k-mers in haplotype {}...\033[0m".format(hap_id), file=sys.stderr)
            paths = files_string.split(",")
            for p in paths:
                if not os.path.isfile(p):
                    raise IOError("no such file: {}".format(p))
            counter = count_library(paths, k, args.capacity or estimate_capacity(paths))
            libraries.append([counter, None, None])
            print("\033[92mComputing and analyzing histogram...\033[0m", file=sys.stderr)
            histogram_path = os.path.join(args.scratch_dir, "haplotype{}.histogram".format(hap_id))
            rows = write_histogram(histogram_path, counter.histogram())
            min_count, max_count = analyze_histogram(rows, histogram_path)
            print("\033[92mUsing counts in range [{},{}].\033[0m".format(min_count, max_count), file=sys.stderr)
            libraries[-1][1:] = [min_count, max_count]
        (counter_a, min_a, max_a), (counter_b, min_b, max_b) = libraries
        print("\033[92mFinding and dumping k-mers unique to haplotype A...\033[0m", file=sys.stderr)
        n_a = counter_a.unique(counter_b, min_a, max_a, os.path.join(args.outpath, "hapA_only_kmers.txt"))
        print("\033[92mFinding and dumping k-mers unique to haplotype B...\033[0m", file=sys.stderr)
        n_b = counter_b.unique(counter_a, min_b, max_b, os.path.join(args.outpath, "hapB_only_kmers.txt"))
    finally:
        for lib in libraries:
            lib[0].close()
    print("\n\n\033[94m# of unique k-mers in haplotype A: {}\033[0m".format(n_a), file=sys.stderr)
    print("\033[94m# of unique k-mers in haplotype B: {}\033[0m".format(n_b), file=sys.stderr)


if __name__ == "__main__":
    main()
