"""What does a miss of a narrow random request really fetch?    Runs the gather calibration in the
shapes the probe kernels use - (32, 2): two lanes x 16 bytes of a 128-byte line, the entry kernels' front; (64, 4): the key
layout's front; (128, 8) and (128, 4): whole lines; plus (64, 1) / (128, 1): one lane reading the line by itself - and a
streaming read, each shape once per process, and prints the lines every launch gathers; run it under
`rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/calib_pmc.py` and join the per-dispatch counter rows with this
JSON by kernel name (tools/calib_pmc_join.py)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from trio_binning_amd._lib import lib, check
foot = int(sys.argv[1]) << 30 if len(sys.argv) > 1 else 32 << 30
res = {"footprint_bytes": foot, "shapes": {}}
for line, lpl in ((32, 2), (64, 4), (128, 8), (128, 4), (64, 1), (128, 1)):
    lps, ms = C.c_double(), C.c_double()
    check(lib.tbk_calib_gather(0, foot, line, lpl, 2, 1 << 27, 2, C.byref(lps), C.byref(ms)))
    res["shapes"][f"tbk_calib_gather_kernel<{line}, {lpl}, 2>"] = {"request_bytes_per_line": line if lpl > 1 else line, "lanes_per_line": lpl, "Glines_per_s": round(lps.value / 1e9, 2), "ms_per_launch": round(ms.value, 3),
                                                                  "lines_per_launch": int(lps.value * ms.value * 1e-3)}
bps = C.c_double()
check(lib.tbk_calib_stream(0, min(foot, 8 << 30), 3, C.byref(bps)))
res["stream"] = {"GBps": round(bps.value / 1e9, 1), "bytes_per_launch": min(foot, 8 << 30)}
print(json.dumps(res))
