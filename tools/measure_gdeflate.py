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
ap.add_argument("--bench", type=int, default=0, help="N > 0: also time N jobs through the encoder's ring (tbk_gzip_bench_device) and the host encoder beside it")
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
rec = {"text_MB": round(len(text) / 1e6, 1), "members": len(pieces), "out_MB": round(out / 1e6, 1), "ratio": round(out / len(text), 4),
       "call_s": [round(t, 4) for t in times], "members_inflate_to_their_text": ok}
if a.bench:
    # the encoder's own line: the ring in its steady state from pinned memory (what the bin writer drives), one job's kernels between
    # HIP events, the host's encoder (tbk_deflate.cpp, the same coder) on every usable thread beside it
    import ctypes as C
    import threading

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    pinned = kmers.pinned_empty((len(text),), np.uint8)
    pinned[:] = np.frombuffer(text, dtype=np.uint8)
    lens = (C.c_uint64 * len(pieces))(*[len(p) for p in pieces])
    ps, ks, ob = C.c_double(), C.c_double(), C.c_uint64()
    check(lib.tbk_gzip_bench_device(0, C.c_void_p(pinned.ctypes.data), lens, len(pieces), a.bench, C.byref(ps), C.byref(ks), C.byref(ob)))
    threads = kmers.host_threads()
    cap = 2 * (1 << 20) + 4096

    def host_worker(idx, stop_at, done):
        buf = C.create_string_buffer(cap)
        n = C.c_size_t()
        while time.perf_counter() < stop_at:
            p = pieces[idx % len(pieces)]
            lib.tbk_gzip_member(p, len(p), buf, cap, C.byref(n))
            done[idx % threads] += len(p)
            idx += threads
    done = [0] * threads
    t0 = time.perf_counter()
    pool = [threading.Thread(target=host_worker, args=(i, t0 + 3.0, done)) for i in range(threads)]
    for t in pool: t.start()
    for t in pool: t.join()
    host_s = time.perf_counter() - t0
    alg = len(text) + ob.value   # the text read once, the members written once
    rec["bench"] = {
        "metric": "GB/s of FASTQ text into gzip members (the bins of classify-by-kmers' default mode, seq.py:132-134)", "unit": "GB/s", "dtype": "u8",
        "value": round(len(text) / ps.value / 1e9, 2), "what": f"{a.bench} jobs of {len(pieces)} members of 1 MiB through the three-deep ring from pinned host memory: H2D, kernels, D2H overlapped",
        "kernels_only_GB_per_s": round(len(text) / ks.value / 1e9, 2), "kernels_ms_per_job": round(ks.value * 1e3, 3), "pipelined_ms_per_job": round(ps.value * 1e3, 3),
        "roofline": {"bound": "pcie", "achieved": round(len(text) / ps.value / 1e9, 2), "peak": 55.0, "unit": "GB/s of text over the link (H2D; the members go back on the other direction)",
                     "frac": round(len(text) / ps.value / 1e9 / 55.0, 3),
                     "hbm": {"algorithmic_bytes_per_job": alg, "kernels_ms": round(ks.value * 1e3, 3), "achieved_GB_per_s": round(alg / ks.value / 1e9, 1), "peak": 8000.0,
                             "frac": round(alg / ks.value / 1e9 / 8000.0, 4), "traffic": "profiles/r06/gdeflate_pmc.json: 138 MB read + 53 MB written by gd_encode_kernel per 134 MB job (1.05 x algorithmic)"}},
        "cpu_baseline": {"value": round(sum(done) / host_s / 1e9, 2), "unit": "GB/s", "cores": threads, "kind": "port", "sample": f"the library's host encoder (tbk_deflate.cpp: the same coder) on {threads} threads for 3 s over the same members"},
    }
print(json.dumps(rec))
