#!/usr/bin/env python3
"""Probe-kernel rate on lists shaped like real trio-binning input (keys in runs of overlapping
k-mers around SNPs, reads with dense hits), as opposed to bench.py's uniform random keys."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trio_binning_amd import kmers
from trio_binning_amd._lib import check, lib

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=100_000_000)
ap.add_argument("--snp", type=float, default=1 / 500)
ap.add_argument("--reads", type=int, default=20000)
ap.add_argument("--read-len", type=int, default=15000)
a = ap.parse_args()
k = 21
rng = np.random.default_rng(7)
t0 = time.time()
genome = rng.integers(0, 4, a.genome, dtype=np.uint8)
def mutate(g, seed):
    r = np.random.default_rng(seed); h = g.copy()
    pos = np.nonzero(r.random(g.size) < a.snp)[0]
    h[pos] = (h[pos] + r.integers(1, 4, pos.size)) % 4
    return h
hap = [mutate(genome, 1), mutate(genome, 2)]
def canon(codes):
    c = codes.astype(np.uint64); n = c.size - k + 1
    f = np.zeros(n, dtype=np.uint64); r = np.zeros(n, dtype=np.uint64)
    for i in range(k):
        f |= c[i:i + n] << np.uint64(2 * i); r |= (np.uint64(3) - c[i:i + n]) << np.uint64(2 * (k - 1 - i))
    return np.minimum(f, r)
ka, kb = np.unique(canon(hap[0])), np.unique(canon(hap[1]))
oa, ob = np.setdiff1d(ka, kb, assume_unique=True), np.setdiff1d(kb, ka, assume_unique=True)
del ka, kb
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
R, L = a.reads, a.read_len
starts = rng.integers(0, a.genome - L, R)
bases = np.empty((R, L), dtype=np.uint8)
for i in range(R):
    bases[i] = lut[hap[i % 2][starts[i]:starts[i] + L]]
err = rng.random(bases.shape) < 0.002
bases[err] = lut[rng.integers(0, 4, int(err.sum()))]
t_gen = time.time() - t0
A, B = kmers.HashSet.from_keys(oa, k), kmers.HashSet.from_keys(ob, k)
res = {"keys_a": int(oa.size), "keys_b": int(ob.size), "gbases": R * L / 1e9, "gen_s": round(t_gen, 1)}
offs = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
flat = np.ascontiguousarray(bases.reshape(-1))
def dalloc(n):
    p = C.c_void_p(); check(lib.tbk_device_alloc(0, n, C.byref(p))); return p.value
d_b, d_o, d_c = dalloc(flat.size + 64), dalloc(offs.nbytes), dalloc(R * 8)
check(lib.tbk_memcpy_h2d(0, C.c_void_p(d_b), flat.ctypes.data, flat.size)); check(lib.tbk_memcpy_h2d(0, C.c_void_p(d_o), offs.ctypes.data, offs.nbytes))
for load in os.environ.get("TBK_LOADS", "0.1,0.0625,0.25").split(","):
    os.environ["TBK_TABLE_LOAD"] = load
    cls = kmers.Classifier(A, B)
    for _ in range(3): cls.classify_device(d_b, d_o, R, flat.size, d_c)
    cls.sync(); cls.kernel_timing(True)
    for _ in range(10): cls.classify_device(d_b, d_o, R, flat.size, d_c)
    n, ms = cls.kernel_timing_read()
    dbg = None
    try:  # debug builds (-DTBK_COUNTERS) count probe-kernel events
        f = lib.tbk_debug_counters
        buf = (C.c_ulonglong * 8)()
        cls.sync(); f(buf, 1)
        cls.classify_device(d_b, d_o, R, flat.size, d_c); cls.sync(); f(buf, 1)
        w = R * (L - k + 1)
        dbg = {"jsteps": buf[0], "careful_jstep_frac": round(buf[1] / max(buf[0], 1), 4), "careful_substeps_per_jstep": round(buf[2] / max(buf[0], 1), 4),
               "walks_per_window": round(buf[3] / w, 5), "lines_per_window": round(buf[4] / 4 / w, 4)}
    except AttributeError:
        pass
    counts = np.zeros((R, 2), dtype=np.int32); check(lib.tbk_memcpy_d2h(0, counts.ctypes.data, C.c_void_p(d_c), counts.nbytes))
    res[f"load_{load}"] = {"kernel_ms": round(ms / n, 3), "gbases_per_s": round(R * L / (ms / n) / 1e6, 1), "hits_per_read": float(counts.sum() / R),
                           "correct_bin_frac": float(((counts[0::2, 0] > counts[0::2, 1]).mean() + (counts[1::2, 1] > counts[1::2, 0]).mean()) / 2), **cls.stats(), **({"counters": dbg} if dbg else {})}
    cls.close()
print(json.dumps(res))
