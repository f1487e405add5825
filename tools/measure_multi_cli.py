#!/usr/bin/env python3
"""classify-by-kmers on a device list against the one-device run, at a size where many batches are
in flight on every ring: same stdout, same three bins (sha256).  On a one-GPU box the list names
device 0 several times (several rings and table replicas on it)."""
import argparse, hashlib, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--kmers", type=int, default=5_000_000)
ap.add_argument("--reads", type=int, default=60_000)
ap.add_argument("--read-len", type=int, default=15_000)
ap.add_argument("--devices", default="0,0,0")
a = ap.parse_args()
k = 21
rng = np.random.default_rng(2)
tmp = tempfile.mkdtemp(prefix="tbk_multi_")
keys = np.unique(rng.integers(0, 4**k, 2 * a.kmers + a.kmers // 20, dtype=np.uint64)); rng.shuffle(keys); keys = keys[: 2 * a.kmers]
def decode(v):
    out = np.empty((v.size, k + 1), dtype=np.uint8); lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(k): out[:, i] = lut[((v >> np.uint64(2 * i)) & np.uint64(3)).astype(np.int64)]
    out[:, k] = 10; return out
la, lb = decode(keys[: a.kmers]), decode(keys[a.kmers:])
fa, fb = os.path.join(tmp, "hapA.txt"), os.path.join(tmp, "hapB.txt"); la.tofile(fa); lb.tofile(fb)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, a.reads * a.read_len)].reshape(a.reads, a.read_len)
for r in range(a.reads):
    src = la if r % 3 == 0 else lb
    for j in range(r % 7):
        p = j * (a.read_len // 8) + 11; bases[r, p:p + k] = src[(r * 7 + j) % a.kmers, :k]
fq = os.path.join(tmp, "reads.fastq")
with open(fq, "wb") as fh:
    for r in range(a.reads):
        fh.write(b"@read%d c\n" % r); fh.write(bases[r].tobytes()); fh.write(b"\n+\n"); fh.write(b"I" * a.read_len); fh.write(b"\n")
res = {"reads": a.reads, "gbases": a.reads * a.read_len / 1e9}
digests = {}
for devices in ("0", a.devices):
    out = os.path.join(tmp, "out_" + devices.replace(",", "_")); os.makedirs(out)
    env = dict(os.environ, PYTHONPATH=ROOT, TBK_STATS="1", TBK_DEVICES=devices, TBK_BATCH_BASES=str(16 << 20))
    t = time.time()
    p = subprocess.run([sys.executable, "-m", "trio_binning_amd.classify_by_kmers", fq, fa, fb, "--no-gzip-output",
                        "--haplotype-a-out-prefix", os.path.join(out, "hapA"), "--haplotype-b-out-prefix", os.path.join(out, "hapB"),
                        "--unclassified-out-prefix", os.path.join(out, "unc")], env=env, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    st = [l for l in p.stderr.decode().splitlines() if l.startswith("tbk-stats ")]
    d = {"stdout": hashlib.sha256(p.stdout).hexdigest()}
    for f in sorted(os.listdir(out)): d[f] = hashlib.sha256(open(os.path.join(out, f), "rb").read()).hexdigest()
    digests[devices] = d
    res[devices] = {"wall_s": round(time.time() - t, 2), "stages": json.loads(st[-1][10:]) if st else None}
res["identical"] = digests["0"] == digests[a.devices]
res["bins"] = {b: sum(1 for l in p.stdout.decode().splitlines() if l.split("\t")[1] == b) for b in "ABU"}
print(json.dumps(res))
