"""Random-line rate by access shape (same-box): which (request bytes, lanes per line, loads in flight) reaches the ceiling?
(128, 8): 8 lanes x 16 B; (128, 4): the whole-line probe, 4 lanes x two 64-byte halves; (64, 4): the front probe, 4 lanes x 16 B;
(32, 2): the entry layout's pair-cooperative probe, 2 lanes x 16 B - 32 lines touched per wave instruction."""
import ctypes as C, sys, json
sys.path.insert(0, '.')
from trio_binning_amd._lib import lib, check
res = {}
for foot in (60 << 30, 16 << 30):
    for line, lpl in ((128, 8), (128, 4), (64, 4), (32, 2)):
        for inf in (1, 2, 4):
            lps, ms = C.c_double(), C.c_double()
            check(lib.tbk_calib_gather(0, foot, line, lpl, inf, 1 << 28, 3, C.byref(lps), C.byref(ms)))
            res[f"{foot >> 30}GB_line{line}_lanes{lpl}_inflight{inf}"] = round(lps.value / 1e9, 2)
print(json.dumps(res))
