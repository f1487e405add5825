#!/usr/bin/env python3
"""End-to-end timing of the find-unique-kmers CLI on synthetic parents (two haplotypes of one random
genome, short reads with errors written as FASTQ), then of classify-by-kmers on long reads of both
haplotypes with the lists it made: the whole pipeline on files."""
import argparse, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=20_000_000)
ap.add_argument("--coverage", type=float, default=30.0)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--split", type=int, default=1, help="files per parent library")
ap.add_argument("--gzip", action="store_true", help="gzip the parents' FASTQ files")
a = ap.parse_args()
k = 21
rng = np.random.default_rng(5)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
base = rng.integers(0, 4, a.genome, dtype=np.uint8)
def mutate():
    h = base.copy(); pos = np.nonzero(rng.random(a.genome) < 1 / 500)[0]
    h[pos] = (h[pos] + rng.integers(1, 4, pos.size)) % 4
    return h
haps = [mutate(), mutate()]
tmp = tempfile.mkdtemp(prefix="tbk_unique_")
import gzip as _gzip
def write_fastq(path, codes2d, prefix):
    n, L = codes2d.shape
    text = lut[codes2d]
    qual = b"I" * L
    with (_gzip.open(path, "wb", compresslevel=1) if path.endswith(".gz") else open(path, "wb")) as fh:
        for i in range(n):
            fh.write(b"@%s%d\n" % (prefix, i)); fh.write(text[i].tobytes()); fh.write(b"\n+\n"); fh.write(qual); fh.write(b"\n")
files = []
n_reads = int(a.genome * a.coverage / a.read_len)
for h, name in zip(haps, ("mother", "father")):
    starts = rng.integers(0, a.genome - a.read_len, n_reads)
    reads = h[starts[:, None] + np.arange(a.read_len)[None, :]]
    err = rng.random(reads.shape) < 0.003
    reads[err] = (reads[err] + rng.integers(1, 4, int(err.sum()))) % 4
    parts = []
    for j in range(a.split):
        path = os.path.join(tmp, "%s_%d.fastq%s" % (name, j, ".gz" if a.gzip else ""))
        write_fastq(path, reads[j * n_reads // a.split:(j + 1) * n_reads // a.split], name.encode()); parts.append(path)
    files.append(",".join(parts))
    del reads
env = dict(os.environ, PYTHONPATH=ROOT)
t = time.time()
p = subprocess.run([sys.executable, "-m", "trio_binning_amd.find_unique_kmers", "-k", str(k), "-o", tmp, "-s", tmp, files[0], files[1]], env=env, capture_output=True)
t_unique = time.time() - t
assert p.returncode == 0, p.stderr.decode()[-2000:]
err = p.stderr.decode()
lists = [os.path.join(tmp, "hapA_only_kmers.txt"), os.path.join(tmp, "hapB_only_kmers.txt")]
n_list = [os.path.getsize(f) // (k + 1) for f in lists]
# offspring long reads: one haplotype each
L, n_long = 15000, 2000
starts = rng.integers(0, a.genome - L, n_long)
long_reads = np.stack([haps[i % 2][s:s + L] for i, s in enumerate(starts)])
fq = os.path.join(tmp, "child.fastq"); write_fastq(fq, long_reads, b"child")
t = time.time()
p2 = subprocess.run([sys.executable, "-m", "trio_binning_amd.classify_by_kmers", fq, lists[0], lists[1], "--no-gzip-output",
                     "--haplotype-a-out-prefix", os.path.join(tmp, "A"), "--haplotype-b-out-prefix", os.path.join(tmp, "B"),
                     "--unclassified-out-prefix", os.path.join(tmp, "U")], env=env, capture_output=True)
t_classify = time.time() - t
assert p2.returncode == 0, p2.stderr.decode()[-2000:]
bins = [l.split("\t")[1] for l in p2.stdout.decode().splitlines()]
right = sum(1 for i, b in enumerate(bins) if b == "AB"[i % 2])
print(json.dumps({"genome": a.genome, "parent_reads": n_reads, "parent_gbases_each": round(n_reads * a.read_len / 1e9, 3),
                  "files_per_parent": a.split, "gzip": a.gzip, "file_GB_each_parent": round(sum(os.path.getsize(f) for f in files[0].split(",")) / 1e9, 2), "find_unique_s": round(t_unique, 2),
                  "cutoffs": [l for l in err.splitlines() if "Using counts" in l], "list_sizes": n_list,
                  "classify_s": round(t_classify, 2), "child_reads": n_long, "binned_to_the_right_parent": right}))
for f in os.listdir(tmp): os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
