#!/usr/bin/env python3
"""Random-line gather rate from cache-sized footprints (L2 4 MB per XCD, Infinity Cache 256 MB):
what a partitioned (sort-then-probe) design could expect from cache-resident table slices."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trio_binning_amd._lib import check, lib
res = {}
for mb in (1, 2, 4, 8, 16, 32, 64, 128, 192, 256, 512, 1024):
    fp = mb << 20
    row = {}
    for line, lpl in ((128, 8), (64, 4)):
        lps, ms = C.c_double(), C.c_double()
        check(lib.tbk_calib_gather(0, fp, line, lpl, 4, 1 << 28, 3, C.byref(lps), C.byref(ms)))
        row[f"line{line}"] = round(lps.value / 1e9, 2)
    res[f"{mb}MB"] = row
    print(mb, row, flush=True)
print(json.dumps(res))
