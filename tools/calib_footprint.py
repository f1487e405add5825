#!/usr/bin/env python3
"""Random-line gather ceiling as a function of footprint (TLB reach / page effects)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trio_binning_amd._lib import check, lib
res = {}
for gb in (1, 4, 16, 38.4, 77, 128, 200):
    fp = int(gb * 1e9)
    row = {}
    for line, lpl in ((128, 8), (64, 4)):
        lps, ms = C.c_double(), C.c_double()
        check(lib.tbk_calib_gather(0, fp, line, lpl, 4, 1 << 28, 3, C.byref(lps), C.byref(ms)))
        row[f"line{line}"] = round(lps.value / 1e9, 2)
    res[f"{gb}GB"] = row
    print(gb, row, flush=True)
bps = C.c_double()
check(lib.tbk_calib_stream(0, 8 << 30, 5, C.byref(bps)))
res["stream_GBps"] = round(bps.value / 1e9, 1)
print(json.dumps(res))
