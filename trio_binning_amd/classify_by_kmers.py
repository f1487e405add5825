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
from typing import Tuple

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
    """The classify pipeline of this run: one feeder thread and stream ring per device of TBK_DEVICES
    (default: every visible device; a device may repeat), tables replicated, batches dealt to them and
    taken back in input order - the library's ``tbk_pipeline``, also when that is a single device."""
    # (how the table is built is an argument of the library - kmers.Options; the command line has no flags for it, as the
    # reference has none, so the TBK_* variables of the environment are its fallback: Options.from_env)
    # A table built by inserts that merge keys (entries, wide entries) is asked for every line of both lists before the library
    # hands it out, on every device (tbk_options.verify_build; c/kmers.c:112-122 stores every line): a wrong table fails here.
    # TBK_VERIFY_BUILD=1 extends that to every layout, =0 switches it off.
    options = kmers.Options.from_env() if _lib.HAS_OPTIONS else None
    classifier = kmers.MultiClassifier(haplotype_a_kmers, haplotype_b_kmers, kmers.visible_devices(), options=options)
    if os.environ.get("TBK_VERIFY_BUILD", "") not in ("", "0") or os.environ.get("TBK_STATS", "") not in ("", "0"):
        print("tbk-verify " + str([classifier._part(i).verified() for i in range(len(classifier.devices))]), file=sys.stderr)
    return classifier


def main():
    """Main method of program"""
    args = parse_args()

    num_a = kmers.get_number_kmers_in_set(args.haplotype_a_kmers)
    num_b = kmers.get_number_kmers_in_set(args.haplotype_b_kmers)
    import time

    t_start = time.perf_counter()
    classifier = make_classifier(args.haplotype_a_kmers, args.haplotype_b_kmers)
    t_built = time.perf_counter()

    # The loop of the reference (classify_by_kmers.py:99-117: count, score, bin, print one read at a time)
    # runs inside the library on native threads - a reader filling batches in pinned memory (their bases
    # packed for the link on the way), this thread feeding the device(s) and taking the batches back in
    # input order, a writer scoring, binning, writing the three bins and the TSV.  Python only names the
    # files; the TSV goes to the process's stdout descriptor.
    names = seq.output_names(args.haplotype_a_out_prefix, args.haplotype_b_out_prefix, args.unclassified_out_prefix,
                             output_extension(args.reads), not args.no_gzip_output)
    # zlib level of the gzip members when zlib is asked for (TBK_GZIP_ENCODER): default 6; the reference's
    # gzip.open uses 9, which only changes the container bytes, never the decompressed bins
    level = int(os.environ.get("TBK_GZIP_LEVEL", "-1"))
    sys.stdout.flush()
    spool = None
    try:
        tsv_fd = sys.stdout.fileno()
    except (AttributeError, OSError, ValueError):  # stdout is not a file (a test harness's capture): spool, then hand over
        import tempfile

        spool = tempfile.TemporaryFile()
        tsv_fd = spool.fileno()
    try:
        stats = classifier.classify_file(args.reads, num_a, num_b, names, not args.no_gzip_output, level, tsv_fd, _BATCH_BASES, _BATCH_READS)
    finally:
        if spool is not None:
            spool.seek(0)
            sys.stdout.write(spool.read().decode())
            spool.close()
    devices = list(getattr(classifier, "devices", [classifier.device]))
    classifier.close()
    if os.environ.get("TBK_STATS"):
        # stderr is free-form in the reference too (progress chatter); stdout stays pure TSV
        import json

        stats["loop_s"] = stats["total_s"]  # the native read / classify / write loop alone
        stats["table_build_s"] = t_built - t_start
        stats["devices"] = devices
        stats["numpy_loaded"] = "numpy" in sys.modules  # (the command line's path needs none of it: kmers imports it at first use)
        stats["total_s"] = time.perf_counter() - t_start
        stats["gbases_per_s"] = stats["bases"] / stats["total_s"] / 1e9 if stats["total_s"] > 0 else 0.0
        print("tbk-stats " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()}), file=sys.stderr)


if __name__ == "__main__":
    main()
