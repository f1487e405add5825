#!/usr/bin/env python3
"""k-mer counting rate of the find-unique-kmers step (tbk_counter_add_device) on synthetic short
reads generated in HBM: reads of one haplotype of an implicit random genome, with substitution
errors, at a given coverage.  Reports Gbases/s with the table's load and the count histogram's
shape (error k-mers at 1, the coverage peak)."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trio_binning_amd import kmers
from trio_binning_amd._lib import check, lib

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=200_000_000)
ap.add_argument("--coverage", type=float, default=20.0)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--error-rate", type=float, default=0.002)
ap.add_argument("--batch-bases", type=int, default=1_000_000_000)
ap.add_argument("-k", type=int, default=21)
ap.add_argument("--dump", default="", help="also time subtract + sorted dump of the k-mers counted 2..255 times to this path")
a = ap.parse_args()
dev, k, L = 0, a.k, a.read_len
R = a.batch_bases // L
n_batches = max(1, int(a.genome * a.coverage / (R * L)))
def dalloc(n):
    p = C.c_void_p(); check(lib.tbk_device_alloc(dev, n, C.byref(p))); return p.value
d_bases, d_offs = dalloc(R * L + 64), dalloc((R + 1) * 8)
err24 = int(a.error_rate * (1 << 24))
capacity = int(a.genome * 1.05 + n_batches * R * L * a.error_rate * k * 1.1) + (1 << 20)
t0 = time.time()
ctr = kmers.KmerCounter(k, capacity)
t_create = time.time() - t0
gen_s = count_s = 0.0
for b in range(n_batches):
    t = time.time()
    check(lib.tbk_synth_hap_reads_device(dev, 0x5EED0001, a.genome, 0, 0x5EED0003, b * R, R, L, err24, C.c_void_p(d_bases), C.c_void_p(d_offs)))
    gen_s += time.time() - t
    t = time.time()
    ctr.add_device(d_bases, d_offs, R, R * L)
    check(lib.tbk_device_sync(dev))
    count_s += time.time() - t
t = time.time(); hist = ctr.histogram(); hist_s = time.time() - t
# parity at scale + CPU datum: the oracle's single-thread C counter on a prefix of the last batch
# against a GPU counter fed the same prefix
from oracle import binding
orc = binding.load()
n_cpu = min(R, max(1000, int(40e6 // L)))
h_bases = np.empty(n_cpu * L, dtype=np.uint8); h_offs = np.arange(n_cpu + 1, dtype=np.uint64) * np.uint64(L)
check(lib.tbk_memcpy_d2h(dev, h_bases.ctypes.data, C.c_void_p(d_bases), h_bases.size))
t = time.time(); cpu_hist = orc.kmer_histogram(h_bases, h_offs, k, n_cpu * L); cpu_s = time.time() - t
with kmers.KmerCounter(k, n_cpu * L) as small:
    small.add(h_bases, h_offs)
    same = bool(np.array_equal(small.histogram(), cpu_hist))
st = ctr.stats()
dump = None
if a.dump:
    with kmers.KmerCounter(k, 1 << 16) as nothing:
        t = time.time(); n_dumped = ctr.unique(nothing, 2, 255, a.dump); dump = {"kmers": n_dumped, "seconds": round(time.time() - t, 2), "file_GB": round(os.path.getsize(a.dump) / 1e9, 2)}
    os.remove(a.dump)
peak = int(np.argmax(hist[3:]) + 3)
print(json.dumps({"k": k, "genome": a.genome, "reads": n_batches * R, "read_len": L, "gbases": n_batches * R * L / 1e9,
                  "count_s": round(count_s, 3), "gbases_per_s": round(n_batches * R * L / count_s / 1e9, 2),
                  "gkmers_per_s": round(n_batches * R * (L - k + 1) / count_s / 1e9, 2),
                  "distinct": int(hist[0]), "singletons": int(hist[1]), "coverage_peak_at": peak, "table_load": round(int(hist[0]) / st["n_slots"], 3),
                  "table_GB": round(st["table_bytes"] / 1e9, 1), "create_s": round(t_create, 2), "histogram_s": round(hist_s, 3),
                  "cpu_baseline": {"kind": "port", "cores": 1, "mbases_per_s": round(n_cpu * L / cpu_s / 1e6, 1),
                                   "sample": f"oracle C counter (one open-addressing table, rolling canonical k-mers) on {n_cpu} reads ({n_cpu * L / 1e6:.0f} Mbases)"},
                  "parity": {"gpu_histogram_equals_cpu": same, "reads_checked": n_cpu}, "dump": dump}))
