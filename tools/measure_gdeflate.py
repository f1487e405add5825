#!/usr/bin/env python3
"""The GPU gzip encoder by itself (csrc/tbk_gdeflate.hip) on FASTQ text shaped like a bin's: a job of 1 MiB members per call of
tbk_gzip_members_device, as the bin writer cuts them.  Prints GB/s of text per call (create + copy in + kernels + copy out, synchronous:
the writer overlaps three jobs) and checks every member with zlib.  Under `rocprofv3 --kernel-trace --stats` this is what
profiles/r06/gdeflate_kernel_stats.csv was taken from.

    python tools/measure_gdeflate.py [--mb 134] [--reps 5] [--qual hifi|const] [--read-len 15000]
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=134)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--qual", default="hifi")
ap.add_argument("--read-len", type=int, default=15000)
a = ap.parse_args()
from trio_binning_amd import seq  # noqa: E402

rng = np.random.default_rng(3)
L = a.read_len
n = max(1, a.mb * 1_000_000 // (2 * L + 20))
recs = []
for i in range(n):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].tobytes()
    if a.qual == "const":
        q = b"I" * L
    else:
        qv = np.clip(rng.normal(60, 15, L), 2, 93).astype(np.uint8)
        qv[rng.random(L) < 0.6] = 93
        q = (qv + 33).tobytes()
    recs.append(b"@read%09d c\n" % i + s + b"\n+\n" + q + b"\n")
text = b"".join(recs)
pieces = [text[i:i + (1 << 20)] for i in range(0, len(text), 1 << 20)]
times = []
for r in range(a.reps):
    t0 = time.perf_counter()
    members = seq.gzip_members_device(pieces)
    times.append(time.perf_counter() - t0)
ok = all(zlib.decompressobj(31).decompress(m) == p for m, p in zip(members, pieces))
out = sum(len(m) for m in members)
print(json.dumps({"text_MB": round(len(text) / 1e6, 1), "members": len(pieces), "out_MB": round(out / 1e6, 1), "ratio": round(out / len(text), 4),
                  "call_s": [round(t, 4) for t in times], "text_GB_per_s_best_call": round(len(text) / min(times) / 1e9, 2), "members_inflate_to_their_text": ok}))
