#!/usr/bin/env python3
"""The LITERAL drop-in's speed: the loop of the reference's driver (classify_by_kmers.py:99-102 - one
`count_kmers_in_read` per read, then score and bin) over INTEGRATION.md's snippet C, i.e. the reference's own
binding (kmers.py:62-86,125-159) bound to libtbk_hip.so's reference-named symbols (tbk_compat.cpp).

    python tools/measure_dropin.py [--reads 10000] [--read-len 15000] [--keys 1000000] [--out gpurun_out/dropin.json]

Synthetic input of this repository's own: two lists of --keys distinct canonical 21-mers written as text (the
reference's input format), --reads reads of --read-len random bases with 30 list k-mers planted in nine of ten.
Reports seconds per call and Gbases/s for the literal loop, and - same reads, same lists - for the batch path
(`trio_binning_amd.kmers.Classifier.classify_batch`) whose counts the loop's must equal.  No oracle and no reference
code is involved: the caller IS the snippet in INTEGRATION.md."""
import argparse
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def snippet(letter):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(# integration-snippet: %s\n.*?)```" % letter, text, flags=re.S)
    assert len(blocks) == 1
    return blocks[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000)
    ap.add_argument("--read-len", type=int, default=15_000)
    ap.add_argument("--keys", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dropin.json"))
    args = ap.parse_args()

    import numpy as np

    from trio_binning_amd import kmers

    rng = np.random.default_rng(20260604)
    k, n = args.k, args.keys
    # distinct canonical k-mers from random codes
    comp = str.maketrans("ACGT", "TGCA")
    seen, lists = set(), ([], [])
    codes = rng.integers(0, 4, size=(int(2.2 * n), k), dtype=np.uint8)
    letters = np.array(list("ACGT"))
    for row in letters[codes]:
        s = "".join(row)
        rc = s.translate(comp)[::-1]
        c = min(s, rc)   # (the packed order of c/kmers.c:50-72 and the lexicographic one pick the same strand: SURVEY appendix A)
        if c in seen:
            continue
        seen.add(c)
        lists[len(seen) % 2].append(c)
        if len(seen) == 2 * n:
            break
    tmp = tempfile.mkdtemp(prefix="tbk_dropin_")
    fa, fb = os.path.join(tmp, "hapA.txt"), os.path.join(tmp, "hapB.txt")
    for path, lst in ((fa, lists[0]), (fb, lists[1])):
        with open(path, "w") as fh:
            fh.write("\n".join(lst) + "\n")
    reads = []
    for r in range(args.reads):
        seq = letters[rng.integers(0, 4, size=args.read_len, dtype=np.uint8)]
        origin = r % 10
        if origin < 9:
            own, other = (lists[0], lists[1]) if origin % 2 == 0 else (lists[1], lists[0])
            for j, pos in enumerate(range(100, args.read_len - k, max(k + 1, (args.read_len - 200) // 33))):
                if j >= 33:
                    break
                src = own if j < 30 else other
                km = src[int(rng.integers(0, len(src)))]
                if rng.integers(0, 2):
                    km = km.translate(comp)[::-1]
                seq[pos:pos + k] = list(km)
        reads.append("".join(seq))
    total = sum(len(s) for s in reads)

    ns = {"__file__": os.path.join(ROOT, "trio_binning_amd", "kmers.py"), "__name__": "reference_kmers_unpatched"}
    exec(compile(snippet("C"), "INTEGRATION.md[C]", "exec"), ns)
    t0 = time.perf_counter()
    a, b = ns["create_kmer_hash_set"](fa), ns["create_kmer_hash_set"](fb)
    num_a, num_b = ns["get_number_kmers_in_set"](a), ns["get_number_kmers_in_set"](b)
    t_lists = time.perf_counter() - t0
    sf_a, sf_b = 1.0 * max(num_a, num_b) / num_a, 1.0 * max(num_a, num_b) / num_b   # classify_by_kmers.py:72-76
    t0 = time.perf_counter()
    ns["count_kmers_in_read"](reads[0], a, b)   # (the first call builds the paired table)
    t_first = time.perf_counter() - t0
    loop_counts, bins = [], {"A": 0, "B": 0, "U": 0}
    per_call = []
    t0 = time.perf_counter()
    for s in reads:                              # classify_by_kmers.py:99-115
        t1 = time.perf_counter()
        ca, cb = ns["count_kmers_in_read"](s, a, b)
        per_call.append(time.perf_counter() - t1)
        sa, sb = ca * sf_a, cb * sf_b
        bins["A" if sa > sb else "B" if sb > sa else "U"] += 1
        loop_counts.append((ca, cb))
    t_loop = time.perf_counter() - t0
    per_call = np.array(per_call)

    ha, hb = kmers.HashSet.from_file(fa, 0), kmers.HashSet.from_file(fb, 0)
    with kmers.Classifier(ha, hb) as cls:
        bases, offs = kmers.pack_reads(reads)
        cls.classify_batch(bases, offs)
        t0 = time.perf_counter()
        batch_counts = cls.classify_batch(bases, offs)
        t_batch = time.perf_counter() - t0
    equal = bool(np.array_equal(np.array(loop_counts, dtype=np.int32), batch_counts))
    rec = {
        "what": "the reference driver's per-read loop (classify_by_kmers.py:99-115) over INTEGRATION.md snippet C: one count_kmers_in_read (tbk_compat.cpp) per read",
        "reads": args.reads, "read_len": args.read_len, "bases": total, "k": k, "keys_per_list": n,
        "lists_loaded_s": round(t_lists, 3), "first_call_s_builds_the_table": round(t_first, 3),
        "loop_s": round(t_loop, 3), "us_per_call_mean": round(1e6 * float(per_call.mean()), 1), "us_per_call_median": round(1e6 * float(np.median(per_call)), 1),
        "us_per_call_p99": round(1e6 * float(np.quantile(per_call, 0.99)), 1), "us_per_call_p999": round(1e6 * float(np.quantile(per_call, 0.999)), 1),
        "us_per_call_max": round(1e6 * float(per_call.max()), 1), "calls_over_1ms": int((per_call > 1e-3).sum()),
        "literal_loop_gbases_per_s": round(total / t_loop / 1e9, 4),
        "batch_path_s": round(t_batch, 4), "batch_path_gbases_per_s": round(total / t_batch / 1e9, 3),
        "counts_equal_the_batch_path": equal, "hits": [int(batch_counts[:, 0].sum()), int(batch_counts[:, 1].sum())], "bins": bins,
        "device": __import__("trio_binning_amd._lib", fromlist=["x"]).device_name(0),
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))
    if not equal:
        raise SystemExit("the literal loop's counts differ from the batch path's")


if __name__ == "__main__":
    main()
