#!/usr/bin/env python3
"""Golden vectors for the find-unique-kmers step, from the REAL reference.

The reference's find_unique_kmers.py is subprocess glue around KMC, which is not installed here;
the one piece of arithmetic that is the reference's own is analyze_histogram's choice of the count
cut-offs (find_unique_kmers.py:106-168).  This script imports the reference module from
/root/reference, replaces subprocess.check_call inside it by a stand-in for
`kmc_tools transform <db> histogram <path>` that writes a prepared histogram to <path>, calls the
reference's analyze_histogram and records (histogram -> cut-offs or HistogramError).  Dev container
only; the fixture is data.

    python tests/golden/make_golden_unique.py   ->  tests/golden/unique_cutoffs.json
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
from trio_binning import find_unique_kmers as ref  # noqa: E402


def run(hist):
    """hist: list of (count, number of k-mers) rows, as kmc_tools writes them"""
    def fake_check_call(cmd):
        assert cmd[1:3] == ["transform", "db"] and cmd[3] == "histogram"
        with open(cmd[4], "w") as fh:
            for c, n in hist:
                fh.write("%d\t%d\n" % (c, n))
    ref.check_call = fake_check_call
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stderr(io.StringIO()) as err:
        try:
            lo, hi = ref.analyze_histogram("db", "kmc", tmp)
            return {"min": lo, "max": hi, "warned": "WARNING" in err.getvalue()}
        except ref.HistogramError:
            return {"error": "HistogramError"}


def bimodal(rng, err0, decay, peak_at, peak_h, width, n=255, floor=0):
    rows = []
    for c in range(1, n + 1):
        e = int(err0 * decay ** (c - 1))
        g = int(peak_h * 2.718281828 ** (-((c - peak_at) ** 2) / (2.0 * width * width)))
        rows.append((c, max(floor, e + g + rng.randint(0, 3))))
    return rows


def main():
    rng = random.Random(20240601)
    cases = []
    # typical: error k-mers decaying from count 1 (KMC's -ci2 default zeroes count 1), coverage peak later
    for _ in range(40):
        h = bimodal(rng, rng.randint(10**5, 10**8), rng.uniform(0.2, 0.8), rng.randint(8, 90), rng.randint(10**4, 10**7),
                    rng.uniform(2, 15), n=rng.choice([60, 120, 255]))
        if rng.random() < 0.7:
            h[0] = (1, 0)  # kmc -ci2: nothing counted once is kept
        cases.append(h)
    # degenerate shapes
    cases.append([(c, 1000 - c) for c in range(1, 200)])                # monotone down: no minimum
    cases.append([(c, c) for c in range(1, 200)])                        # monotone up
    cases.append([(1, 0), (2, 50), (3, 40), (4, 45), (5, 44), (6, 30)])  # tiny, cut-offs close together
    cases.append([(1, 0), (2, 5), (3, 9), (4, 20), (5, 3)])              # rises right after count 2
    cases.append([(1, 7), (2, 5), (3, 5), (4, 5), (5, 6), (6, 4)])       # plateau
    cases.append([(1, 0), (2, 0), (3, 0), (4, 0)])                       # empty database
    cases.append([(1, 100)])                                             # single row
    cases.append([])                                                     # empty file
    for _ in range(30):                                                  # noise
        n = rng.randint(3, 80)
        cases.append([(c, rng.randint(0, 1000)) for c in range(1, n + 1)])
    out = [{"histogram": h, "expect": run(h)} for h in cases]
    with open(os.path.join(HERE, "unique_cutoffs.json"), "w") as fh:
        json.dump({"source": "find_unique_kmers.py:106-168 (analyze_histogram) of the reference, check_call replaced", "cases": out}, fh)
    print(len(out), "cases;", sum(1 for c in out if "error" in c["expect"]), "HistogramError;",
          sum(1 for c in out if c["expect"].get("warned")), "warned")


if __name__ == "__main__":
    main()
