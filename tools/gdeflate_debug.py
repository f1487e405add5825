import sys, zlib, gzip
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from trio_binning_amd import seq
import test_gpu_deflate as T
rng = np.random.default_rng(11)
pieces = [
    T.fastq(rng, 40, 15000, "hifi"), T.fastq(rng, 40, 15000, "const"), T.fastq(rng, 3000, (50, 300), "binned"), T.fastq(rng, 3, 200_000, "hifi"),
    b"", b"A", b"\n", b"AB", b"A" * 100_000, bytes(range(256)) * 300,
    rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(), bytes(rng.integers(0, 2, 70_000, dtype=np.uint8) * 255),
    (b">r\n" + b"ACGT" * 20 + b"\n") * 2000, b"".join(bytes([65 + (i % 7)]) * (i % 300 + 1) for i in range(2000)),
]
members = seq.gzip_members_device(pieces)
for i, (p, m) in enumerate(zip(pieces, members)):
    try:
        d = zlib.decompressobj(31).decompress(m)
        ok = d == p
        first = next((j for j in range(min(len(d), len(p))) if d[j] != p[j]), None)
        print(i, len(p), len(m), "ok" if ok else f"DIFF at {first} (got {len(d)} bytes)")
    except Exception as e:
        d = zlib.decompressobj(31)
        out = b""
        try:
            for j in range(0, len(m), 64):
                out += d.decompress(m[j:j + 64])
        except Exception as e2:
            pass
        first = next((j for j in range(min(len(out), len(p))) if out[j] != p[j]), len(out))
        print(i, len(p), len(m), "ERROR", e, "good prefix", first)
