#!/usr/bin/env python3
"""End-to-end CLI timing on synthetic files (list parse + table build + read + classify + write).
Not bench.py's metric: it includes file I/O on the host.  DESIGN.md quotes it."""
import argparse, gzip, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--kmers", type=int, default=10_000_000)
ap.add_argument("--reads", type=int, default=20_000)
ap.add_argument("--read-len", type=int, default=15_000)
ap.add_argument("--gz-input", action="store_true", help="gzip the reads file (one ordinary gzip stream)")
a = ap.parse_args()
k = 21
rng = np.random.default_rng(1)
tmp = tempfile.mkdtemp(prefix="tbk_cli_")
keys = np.unique(rng.integers(0, 4**k, 2 * a.kmers + a.kmers // 20, dtype=np.uint64))
rng.shuffle(keys)
keys = keys[: 2 * a.kmers]
def decode(v):
    out = np.empty((v.size, k + 1), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(k):
        out[:, i] = lut[((v >> np.uint64(2 * i)) & np.uint64(3)).astype(np.int64)]
    out[:, k] = 10
    return out
la, lb = decode(keys[: a.kmers]), decode(keys[a.kmers:])
fa, fb = os.path.join(tmp, "hapA.txt"), os.path.join(tmp, "hapB.txt")
la.tofile(fa); lb.tofile(fb)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, a.reads * a.read_len)].reshape(a.reads, a.read_len)
for r in range(a.reads):  # plant 20 list k-mers per read
    src = la if r % 2 == 0 else lb
    for j in range(20):
        p = j * (a.read_len // 20) + 5
        bases[r, p:p + k] = src[(r * 20 + j) % a.kmers, :k]
fq = os.path.join(tmp, "reads.fastq" + (".gz" if a.gz_input else ""))
with (gzip.open(fq, "wb", compresslevel=1) if a.gz_input else open(fq, "wb")) as fh:
    qual = b"I" * a.read_len
    for r in range(a.reads):
        fh.write(b"@read%d some comment\n" % r); fh.write(bases[r].tobytes()); fh.write(b"\n+\n"); fh.write(qual); fh.write(b"\n")
res = {"kmers_per_list": a.kmers, "reads": a.reads, "gbases": a.reads * a.read_len / 1e9, "fastq_GB": os.path.getsize(fq) / 1e9}
env = dict(os.environ, PYTHONPATH=ROOT, TBK_STATS="1")
for mode, extra in (("gzip", []), ("plain", ["--no-gzip-output"])):
    out = os.path.join(tmp, mode); os.makedirs(out)
    t = time.time()
    p = subprocess.run([sys.executable, "-m", "trio_binning_amd.classify_by_kmers", fq, fa, fb,
                        "--haplotype-a-out-prefix", os.path.join(out, "hapA"), "--haplotype-b-out-prefix", os.path.join(out, "hapB"),
                        "--unclassified-out-prefix", os.path.join(out, "unc")] + extra, env=env, capture_output=True)
    dt = time.time() - t
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().splitlines()
    bins = {b: sum(1 for l in lines if l.split("\t")[1] == b) for b in "ABU"}
    st = [l for l in p.stderr.decode().splitlines() if l.startswith("tbk-stats ")]
    res[mode] = {"stages": json.loads(st[-1][10:]) if st else None, "wall_s": round(dt, 2), "gbases_per_s": round(res["gbases"] / dt, 3), "bins": bins,
                 "out_bytes": sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))}
print(json.dumps(res))
