#!/usr/bin/env python3
"""Dev-container only: time the oracle restatement against the real reference build
(oracle/_ref) on the same tables and reads, to relate the on-box CPU baseline (the oracle)
back to the reference itself.  Results are recorded in BASELINE.md / DESIGN.md."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = 21
orc, ref = oracle.load(), oracle.load_ref()
rng = np.random.default_rng(1)
keys = np.unique(rng.integers(0, 4**k, 2 * n + n // 50, dtype=np.uint64))[: 2 * n]
rng.shuffle(keys)
def decode(v):
    out = np.empty((v.size, k + 1), dtype=np.uint8)
    for i in range(k):
        out[:, i] = np.frombuffer(b"ACGT", dtype=np.uint8)[((v >> np.uint64(2 * i)) & np.uint64(3)).astype(np.int64)]
    out[:, k] = 10
    return out.tobytes()
tmp = tempfile.mkdtemp()
fa, fb = os.path.join(tmp, "a.txt"), os.path.join(tmp, "b.txt")
open(fa, "wb").write(decode(keys[:n])); open(fb, "wb").write(decode(keys[n:]))
t = time.time(); ra, rb = ref.create_kmer_hash_set(fa), ref.create_kmer_hash_set(fb); t_ref_build = time.time() - t
t = time.time(); oa, ob = orc.table_from_file(fa), orc.table_from_file(fb); t_orc_build = time.time() - t
reads = ["".join("ACGT"[c] for c in rng.integers(0, 4, 15000)) for _ in range(200)]
t = time.time(); cr = [ref.count_kmers_in_read(r, ra, rb) for r in reads]; t_ref = time.time() - t
t = time.time(); co = [orc.count_kmers_in_read(r, oa, ob) for r in reads]; t_orc = time.time() - t
assert cr == co
mb = len(reads) * 15000 / 1e6
print(f"n={n} per list: reference {mb / t_ref:.3f} Mbases/s, oracle {mb / t_orc:.3f} Mbases/s (ratio {t_ref / t_orc:.2f}); "
      f"table build ref {t_ref_build:.1f}s oracle {t_orc_build:.1f}s")
