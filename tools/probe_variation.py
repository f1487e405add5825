#!/usr/bin/env python3
"""Where does the run-to-run spread of the probe kernel come from?  One process: the same keys and the same
resident reads, the paired table built N times (freed in between), the single-read kernel timed after each
build.  If the time moves with every rebuild the table's placement in HBM is what varies; if it is one number
per process something process-wide (clocks, the runtime's queues) is."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trio_binning_amd import kmers  # noqa: E402
from trio_binning_amd._lib import check, lib  # noqa: E402

dev, k, n, L, R = 0, 21, 300_000_000, 15_000, 262_144


def dalloc(nbytes):
    p = C.c_void_p()
    check(lib.tbk_device_alloc(dev, nbytes, C.byref(p)))
    return p.value


d_keys = dalloc(2 * n * 8)
check(lib.tbk_synth_keys_device(dev, 0x5EED0001, 0, 2 * n, k, C.c_void_p(d_keys)))
a = kmers.HashSet.from_device_keys(d_keys, n, k, device=dev)
b = kmers.HashSet.from_device_keys(d_keys + n * 8, n, k, device=dev)
d_bases, d_offs = dalloc(R * L + 64), dalloc((R + 1) * 8)
check(lib.tbk_synth_reads_device(dev, 0x5EED0002, 0, R, L, 0x5EED0001, n, n, k, 30, 3, C.c_void_p(d_bases), C.c_void_p(d_offs)))
counts = kmers.pinned_empty((R, 2), np.int32)
out = []
hold = []
aligns = [int(x) for x in os.environ.get("VARIATION_ALIGNS", "0").split(",")]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if rep % 2 == 1 and os.environ.get("VARIATION_SHIFT", "1") == "1":  # every other build behind an extra allocation that shifts where the table lands
        hold.append(dalloc((3 << 30) + rep * (1 << 20)))
    os.environ["TBK_TABLE_ALIGN"] = str(aligns[rep % len(aligns)])
    cls = kmers.Classifier(a, b)
    calib = [round(cls.calibrate() / 1e9, 2) for _ in range(3)]
    pairs = round(max(cls.calibrate_pairs(4, 8) for _ in range(2)) / 1e9, 2) if hasattr(cls, "calibrate_pairs") else None
    for _ in range(2):
        cls.wait(cls.submit_device(d_bases, d_offs, R, R * L, counts))
    cls.kernel_timing(True)
    t = time.perf_counter()
    for _ in range(6):
        cls.wait(cls.submit_device(d_bases, d_offs, R, R * L, counts))
    dt = (time.perf_counter() - t) / 6
    nl, ms, single = cls.kernel_timing_read2()
    out.append({"build": rep, "gather_pairs_Glines_per_s": pairs, "single_ms_x_gather": round(single / nl * pairs, 1) if pairs else None, "align": aligns[rep % len(aligns)], "gather_Glines_per_s": calib, "single_ms": round(single / nl, 3), "probe_ms": round(ms / nl, 3), "step_ms": round(dt * 1e3, 3), "sum_a": int(counts[:, 0].sum())})
    print(out[-1], flush=True)
    cls.close()
print(json.dumps(out))
