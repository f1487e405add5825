#!/usr/bin/env python3
"""Native FASTQ reader rate on plain, gzip'ed and bgzip'ed copies of one synthetic file
(text GB/s): the host-side ceiling of both CLIs."""
import argparse, gzip, json, os, struct, sys, tempfile, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trio_binning_amd import seq
from trio_binning_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=60000)
ap.add_argument("--read-len", type=int, default=10000)
ap.add_argument("--qual", choices=["const", "hifi"], default="hifi", help="quality strings: one symbol, or HiFi-like (60 % at the cap, the rest spread)")
a = ap.parse_args()
rng = np.random.default_rng(0)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
def qual():
    if a.qual == "const":
        return b"I" * a.read_len
    qv = np.clip(rng.normal(60, 15, a.read_len), 2, 93).astype(np.uint8)
    qv[rng.random(a.read_len) < 0.6] = 93
    return (qv + 33).tobytes()
text = b"".join(b"@r%d\n" % i + lut[rng.integers(0, 4, a.read_len)].tobytes() + b"\n+\n" + qual() + b"\n" for i in range(a.reads))
def bgzf(data, block=60000):
    out = bytearray()
    for i in range(0, len(data), block):
        piece = data[i:i + block]; c = zlib.compressobj(1, zlib.DEFLATED, -15); body = c.compress(piece) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)
tmp = tempfile.mkdtemp(prefix="tbk_reader_")
files = {"plain": os.path.join(tmp, "t.fastq"), "gzip": os.path.join(tmp, "t_gz.fastq.gz"), "bgzf": os.path.join(tmp, "t_bgzf.fastq.gz")}
open(files["plain"], "wb").write(text); open(files["gzip"], "wb").write(gzip.compress(text, 6)); open(files["bgzf"], "wb").write(bgzf(text))
shared = seq.Batch()   # one batch object for every run: its pinned buffer is allocated (and first touched) once
def run(path):
    t = time.time(); n = 0
    with seq.BatchReader(path) as r:
        while True:
            k = r.next_batch(shared, 256 << 20, 1 << 20)
            if not k: break
            n += k
    assert n == a.reads
    return time.time() - t
res = {"qualities": a.qual, "inflate": os.environ.get("TBK_INFLATE", "own"), "text_GB": round(len(text) / 1e9, 2), "gzip_GB": round(os.path.getsize(files["gzip"]) / 1e9, 2), "host_threads": int(lib.tbk_host_threads())}
# first pass: cold (the batch's pinned buffer is allocated and touched for the first time, as in a short
# run); later passes: the steady state of a long run
for name in ("plain", "gzip", "bgzf"):
    dt = run(files[name]); res[name + "_first_pass"] = {"seconds": round(dt, 3), "text_GB_per_s": round(len(text) / dt / 1e9, 2)}
for name in ("plain", "gzip", "bgzf", "plain"):
    dt = run(files[name]); res[name] = {"seconds": round(dt, 3), "text_GB_per_s": round(len(text) / dt / 1e9, 2)}
for f in files.values(): os.remove(f)
os.rmdir(tmp)
print(json.dumps(res))
