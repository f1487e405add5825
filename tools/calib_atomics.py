#!/usr/bin/env python3
"""Ceiling of fire-and-forget 32-bit atomic adds (what bounds the k-mer counting kernel): random
lines of a table-sized buffer, `run` consecutive adds per line."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trio_binning_amd._lib import check, lib
res = {}
for gb in (0.004, 1, 10, 40):
    row = {}
    for run in (1, 2, 3, 4, 8):
        r = C.c_double()
        check(lib.tbk_calib_atomics(0, int(gb * 1e9), run, 3, C.byref(r)))
        row[f"run{run}"] = round(r.value / 1e9, 1)
    res[f"{gb}GB"] = row
    print(gb, row, flush=True)
print(json.dumps(res))
