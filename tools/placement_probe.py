#!/usr/bin/env python3
"""Is the rate of random lines a property of WHERE 33 GB lie in HBM?  N buffers of the table's size alive side by side, the
probe's gather (two lanes x 16 bytes of a line, 4 lines in flight per pair) timed over each in turn, several rounds: a buffer
that is fast in every round is a placement that is fast; rates that wander together are the device's state."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trio_binning_amd._lib import check, lib  # noqa: E402

dev = 0
n_buf = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
size = int(float(sys.argv[3])) if len(sys.argv) > 3 else 33_424_860_800
lib.tbk_launch_gather.restype = C.c_int
lib.tbk_launch_gather.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
lib.tbk_launch_fill.restype = C.c_int
lib.tbk_launch_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]


def dalloc(nbytes):
    p = C.c_void_p()
    check(lib.tbk_device_alloc(dev, nbytes, C.byref(p)))
    return p.value


sink = dalloc(64)
bufs, made = [], []
for i in range(n_buf):
    t = time.perf_counter()
    bufs.append(dalloc(size))
    made.append(round(time.perf_counter() - t, 3))
    if len(sys.argv) > 4 and sys.argv[4] == "fill":
        assert lib.tbk_launch_fill(bufs[-1], size, 1, None) == 0
check(lib.tbk_device_sync(dev))
done = C.c_uint64()
n_lines = 1 << 28  # ~5.5 ms
rates = [[] for _ in bufs]
for r in range(rounds):
    for i, b in enumerate(bufs):
        assert lib.tbk_launch_gather(b, size & ~255, 32, 2, 4, n_lines >> 3, 5 + r, sink, C.byref(done), None) == 0
        check(lib.tbk_device_sync(dev))
        t = time.perf_counter()
        assert lib.tbk_launch_gather(b, size & ~255, 32, 2, 4, n_lines, 9 + r, sink, C.byref(done), None) == 0
        check(lib.tbk_device_sync(dev))
        rates[i].append(round(done.value / (time.perf_counter() - t) / 1e9, 2))
print(json.dumps({"buffers": n_buf, "bytes": size, "alloc_s": made, "addresses": [hex(b) for b in bufs], "Glines_per_s_by_buffer_and_round": rates}))
