"""Classify reads into bins based on kmers.

This is a script for classifying sequence reads into parental bins
based on the presence of k-mers.
"""
# The module docstring above is the CLI description the reference prints for --help
# (classify_by_kmers.py:1-5,17; asserted by its tests/test_classify_by_kmers.py:16) and is
# kept word for word because it is user-visible output.
#
# Host driver of the MI355X path.  Same command line, defaults, stdout TSV and bin files
# as the reference driver (src/trio_binning/classify_by_kmers.py:14-117); what changes is
# the loop: instead of one ctypes call per read (:99-102) the native reader fills batches
# (pinned memory, C-ABI layout), they stream through the HIP classifier with the next batch's
# copy overlapping the current batch's kernel, and each batch is scored, binned and written by
# the native writer in input order.

import argparse
import os
import sys
from os import path
from typing import List, Tuple

from . import _lib

_lib.warm_up()  # the HIP runtime starts beside the imports and the argument parsing below

from . import kmers, seq  # noqa: E402

# bases per batch handed to the GPU; 3 batches may be in flight
_BATCH_BASES = int(os.environ.get("TBK_BATCH_BASES", str(64 << 20)))  # small enough that pinning the batch buffers is not what a short run waits for
_BATCH_READS = int(os.environ.get("TBK_BATCH_READS", str(1 << 20)))


def parse_args():
    """Parse arguments (same positionals, options, defaults and help as the reference,
    classify_by_kmers.py:14-54; the k-mer tables are built by the ``type=`` callbacks)."""
    parser = argparse.ArgumentParser(
        description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter
    )
    parser.add_argument(
        "reads",
        help="reads to classify into bins, in fasta/q format. Can be gzipped.",
    )
    parser.add_argument(
        "haplotype_a_kmers",
        type=kmers.create_kmer_hash_set,
        help="a list of k-mers unique to haplotype A, one per line",
    )
    parser.add_argument(
        "haplotype_b_kmers",
        type=kmers.create_kmer_hash_set,
        help="a list of k-mers unique to haplotype B, one per line",
    )
    parser.add_argument("--haplotype-a-out-prefix", default="hapA", help="prefix for haplotype A output file")
    parser.add_argument("--haplotype-b-out-prefix", default="hapB", help="prefix for haplotype B output file")
    parser.add_argument("--unclassified-out-prefix", default="unclassified", help="prefix for unclassified output file")
    parser.add_argument("--no-gzip-output", action="store_true", default=False, help="don't gzip the output")
    return parser.parse_args()


def calculate_scaling_factors(haplotype_a_kmers: kmers.HashSet, haplotype_b_kmers: kmers.HashSet) -> Tuple[float, float]:
    """Scaling factors for the k-mer scores (reference classify_by_kmers.py:57-77):
    each count is multiplied by max(nA, nB) / n of its own list, in float64."""
    num_kmers_a = kmers.get_number_kmers_in_set(haplotype_a_kmers)
    num_kmers_b = kmers.get_number_kmers_in_set(haplotype_b_kmers)
    max_num_kmers = max(num_kmers_a, num_kmers_b)
    return 1.0 * max_num_kmers / num_kmers_a, 1.0 * max_num_kmers / num_kmers_b


def output_extension(reads_path: str) -> str:
    """Extension of the bin files.  The reference computes
    ``splitext(reads.rstrip(".gz"))[1]`` (classify_by_kmers.py:90): ``rstrip`` strips the
    character set {'.', 'g', 'z'}, not the suffix, and that quirk decides file names."""
    return path.splitext(reads_path.rstrip(".gz"))[1]


def make_classifier(haplotype_a_kmers, haplotype_b_kmers):
    """The batch classifier for this run: one per device of TBK_DEVICES (default: every visible
    device), tables replicated, batches dealt to them and results taken back in input order; a plain
    ``Classifier`` when that is a single device."""
    devices = kmers.visible_devices()
    if len(devices) > 1:
        return kmers.MultiClassifier(haplotype_a_kmers, haplotype_b_kmers, devices)
    return kmers.Classifier(haplotype_a_kmers, haplotype_b_kmers)


def main():
    """Main method of program"""
    args = parse_args()

    num_a = kmers.get_number_kmers_in_set(args.haplotype_a_kmers)
    num_b = kmers.get_number_kmers_in_set(args.haplotype_b_kmers)
    import time

    stats = {"reads": 0, "bases": 0, "batches": 0, "read_s": 0.0, "gpu_wait_s": 0.0, "write_s": 0.0}
    t_start = time.perf_counter()
    classifier = make_classifier(args.haplotype_a_kmers, args.haplotype_b_kmers)
    stats["table_build_s"] = time.perf_counter() - t_start
    stats["devices"] = list(getattr(classifier, "devices", [classifier.device]))

    # native reader / writer (same records as seq.readfq, same bytes as Read.print)
    reader = seq.BatchReader(args.reads)
    writer = seq.BinWriter(
        args.haplotype_a_out_prefix,
        args.haplotype_b_out_prefix,
        args.unclassified_out_prefix,
        output_extension(args.reads),
        not args.no_gzip_output,
        # zlib level of the gzip members: default 6; the reference's gzip.open uses 9, which only
        # changes the container bytes (and costs 3x the CPU time), never the decompressed bins
        level=int(os.environ.get("TBK_GZIP_LEVEL", "-1")),
    )
    stdout = sys.stdout

    def emit(batch: seq.Batch, counts) -> None:
        """Score, bin and write one batch in input order (classify_by_kmers.py:104-117)."""
        t = time.perf_counter()
        score_a, score_b, bins = kmers.score_and_bin(counts, num_a, num_b)
        writer.write(batch, bins)
        stdout.write(seq.format_tsv(batch, bins, score_a, score_b))
        stats["write_s"] += time.perf_counter() - t

    # Three stages run side by side, one batch apiece (the native calls release the GIL):
    #   reader thread   parses the next batch into pinned memory,
    #   this thread     keeps up to `depth` batches in flight on the GPU,
    #   writer thread   scores, bins and writes finished batches in input order.
    import queue
    import threading

    depth = classifier.depth
    n_batches = depth + 3  # in flight on the GPU(s) + one apiece for reader, queues and writer
    free_q: "queue.Queue" = queue.Queue()
    filled_q: "queue.Queue" = queue.Queue(maxsize=2)
    done_q: "queue.Queue" = queue.Queue(maxsize=2)
    batches = [seq.Batch() for _ in range(n_batches)]
    for b in batches:
        free_q.put(b)
    failure: List[BaseException] = []

    def read_loop() -> None:
        try:
            while not failure:
                batch = free_q.get()
                t = time.perf_counter()
                n = reader.next_batch(batch, _BATCH_BASES, _BATCH_READS)
                stats["read_s"] += time.perf_counter() - t
                if n == 0:
                    free_q.put(batch)
                    break
                stats["reads"] += n
                stats["batches"] += 1
                stats["bases"] += int(batch.arrays()[1][-1])
                filled_q.put(batch)
        except BaseException as exc:  # handed to the main thread
            failure.append(exc)
        finally:
            filled_q.put(None)

    def write_loop() -> None:
        try:
            while True:
                item = done_q.get()
                if item is None:
                    break
                batch, counts = item
                if not failure:
                    emit(batch, counts)
                free_q.put(batch)
        except BaseException as exc:
            failure.append(exc)
            while done_q.get() is not None:  # keep the main thread from blocking on a full queue
                pass

    rt = threading.Thread(target=read_loop, name="tbk-reader", daemon=True)
    wt = threading.Thread(target=write_loop, name="tbk-writer", daemon=True)
    rt.start()
    wt.start()
    in_flight: List[Tuple[int, seq.Batch]] = []  # (ticket, batch) in submission order

    def drain(keep: int) -> None:
        while len(in_flight) > keep:
            ticket, batch = in_flight.pop(0)
            t = time.perf_counter()
            counts = classifier.wait(ticket)
            stats["gpu_wait_s"] += time.perf_counter() - t
            done_q.put((batch, counts))

    try:
        while not failure:
            batch = filled_q.get()
            if batch is None:
                break
            drain(depth - 1)
            in_flight.append((classifier.submit_batch(batch), batch))
        drain(0)
    finally:
        done_q.put(None)
        wt.join()
        if failure:  # unblock a reader waiting for a free batch, then report
            for _ in range(n_batches):
                free_q.put(seq.Batch())
        rt.join(timeout=5)
    if failure:
        raise failure[0]
    free = batches

    # The reference never closes its outputs (interpreter shutdown does); closing here
    # finalises the files at the same point in the byte stream.
    writer.close()
    reader.close()
    for b in free:
        b.close()
    classifier.close()
    if os.environ.get("TBK_STATS"):
        # stderr is free-form in the reference too (progress chatter); stdout stays pure TSV
        import json

        stats["total_s"] = time.perf_counter() - t_start
        stats["gbases_per_s"] = stats["bases"] / stats["total_s"] / 1e9 if stats["total_s"] > 0 else 0.0
        print("tbk-stats " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()}), file=sys.stderr)


if __name__ == "__main__":
    main()
