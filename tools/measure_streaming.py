#!/usr/bin/env python3
"""PCIe-inclusive classify rate: host-resident batches (pinned memory) pushed through
tbk_stream_submit / tbk_stream_wait (H2D on the side stream overlapped with the kernel).
This is NOT bench.py's `value` (that starts with inputs resident in HBM); DESIGN.md quotes it."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trio_binning_amd import _lib, kmers
from trio_binning_amd._lib import check, lib

ap = argparse.ArgumentParser()
ap.add_argument("--kmers-per-list", type=int, default=300_000_000)
ap.add_argument("--reads", type=int, default=16384)
ap.add_argument("--read-len", type=int, default=15000)
ap.add_argument("--batches", type=int, default=24)
ap.add_argument("--pinned", type=int, default=1)
a = ap.parse_args()
dev, k, n = 0, 21, a.kmers_per_list
def dalloc(nb):
    p = C.c_void_p(); check(lib.tbk_device_alloc(dev, nb, C.byref(p))); return p.value
d_keys = dalloc(2 * n * 8)
check(lib.tbk_synth_keys_device(dev, 0x5EED0001, 0, 2 * n, k, C.c_void_p(d_keys)))
A = kmers.HashSet.from_device_keys(d_keys, n, k); B = kmers.HashSet.from_device_keys(d_keys + n * 8, n, k)
cls = kmers.Classifier(A, B)
R, L = a.reads, a.read_len
total = R * L
d_bases, d_offs = dalloc(total + 32), dalloc((R + 1) * 8)
host = []
for b in range(3):
    check(lib.tbk_synth_reads_device(dev, 0x5EED0002, b * R, R, L, 0x5EED0001, n, n, k, 30, 3, C.c_void_p(d_bases), C.c_void_p(d_offs)))
    hb = kmers.pinned_empty((total,), np.uint8) if a.pinned else np.empty(total, dtype=np.uint8)
    ho = kmers.pinned_empty((R + 1,), np.uint64) if a.pinned else np.empty(R + 1, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(dev, hb.ctypes.data, C.c_void_p(d_bases), total))
    check(lib.tbk_memcpy_d2h(dev, ho.ctypes.data, C.c_void_p(d_offs), ho.nbytes))
    host.append((hb, ho))
def run(nb):
    pend = []
    for i in range(nb):
        if len(pend) == cls.depth: cls.wait(pend.pop(0))
        hb, ho = host[i % 3]
        pend.append(cls.submit(hb, ho))
    while pend: cls.wait(pend.pop(0))
run(3)
t = time.perf_counter(); run(a.batches); dt = time.perf_counter() - t
print(json.dumps({"pcie_inclusive_gbases_per_s": round(a.batches * total / dt / 1e9, 2), "batch_gbases": total / 1e9,
                  "batches": a.batches, "pinned_input": bool(a.pinned), "h2d_GBps": round(a.batches * total / dt / 1e9, 2)}))
