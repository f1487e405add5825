#!/usr/bin/env python3
"""The reader's BGZF path at size: a FASTQ of --mb megabytes (HiFi-like qualities) written as BGZF, read through seq.BatchReader on the
host's threads and on the GPU (tbk_fastx_set_device); the two must give the same records (a checksum over names, bases, qualities) and
the seconds of each are printed."""
import argparse, faulthandler, hashlib, os, struct, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=1500)
ap.add_argument("--dir", default="/dev/shm")
a = ap.parse_args()
from trio_binning_amd import seq  # noqa: E402

rng = np.random.default_rng(5)
L = 15000
n = a.mb * 1_000_000 // (2 * L + 30)
path = os.path.join(a.dir, "tbk_bgzf_check.fastq.gz")


def blocks(piece):
    out = bytearray()
    for i in range(0, len(piece), 60000):
        blk = piece[i:i + 60000]
        c = zlib.compressobj(4, zlib.DEFLATED, -15)
        body = c.compress(blk) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk))
    return bytes(out)


with open(path, "wb") as fh, ThreadPoolExecutor(16) as pool:
    for first in range(0, n, 2000):
        m = min(2000, n - first)
        bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (m, L))]
        qv = np.clip(rng.normal(60, 15, (m, L)), 2, 93).astype(np.uint8)
        qv[rng.random((m, L)) < 0.6] = 93
        qv += 33
        text = b"".join(b"@read%09d c\n" % (first + i) + bases[i].tobytes() + b"\n+\n" + qv[i].tobytes() + b"\n" for i in range(m))
        pieces = [text[i:i + 6_000_000] for i in range(0, len(text), 6_000_000)]
        for out in pool.map(blocks, pieces):
            fh.write(out)
    fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
print("file", os.path.getsize(path) / 1e6, "MB;", n, "reads", flush=True)
res = {}
for label, kw in (("gpu", {"device": 0}), ("host", {})):
    h = hashlib.sha256()
    t0 = time.perf_counter()
    count = 0
    with seq.BatchReader(path, **kw) as r:
        print(label, "inflates_on_device", r.inflates_on_device, flush=True)
        b = seq.Batch()
        while r.next_batch(b, 64 << 20, 0):
            count += b.n_reads
            for arr in b.arrays():
                h.update(arr)
    res[label] = (count, h.hexdigest(), round(time.perf_counter() - t0, 3))
    print(label, res[label], flush=True)
assert res["gpu"][:2] == res["host"][:2]
os.remove(path)
