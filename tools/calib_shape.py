import ctypes as C, sys, json
sys.path.insert(0,'.')
from trio_binning_amd._lib import lib, check
res={}
for foot in (60<<30,):
    for line,lpl in ((128,8),(128,4),(64,4)):
        for inf in (1,2,4):
            lps, ms = C.c_double(), C.c_double()
            check(lib.tbk_calib_gather(0, foot, line, lpl, inf, 1<<28, 3, C.byref(lps), C.byref(ms)))
            res[f"line{line}_lanes{lpl}_inflight{inf}"]=round(lps.value/1e9,2)
print(json.dumps(res))
