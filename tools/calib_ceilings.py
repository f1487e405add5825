"""The two ceilings bench.py prices the probe kernel against, in tuned shapes (random_line_frac and
traffic_frac_of_measured_stream were above 1 - the yardsticks sat below what they measured):
  random lines   tbk_calib_gather_pairs - the entry kernels' own request shape (one-wave blocks, two lanes x 16 bytes of a line)
                 over loads in flight per pair x resident waves per SIMD, at the tables' footprints;
  streaming      tbk_calib_stream_nt over unroll x grid size.
Prints one JSON object; the best of each is what profiles/calibration.json records."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from trio_binning_amd._lib import lib, check

res = {"random_lines_Glines_per_s": {}, "stream_GBps": {}}
for foot_gb in (33, 19, 128):
    row = {}
    for waves in (8, 6, 4):
        for inf in (1, 2, 3, 4, 6, 8):
            lps = C.c_double()
            check(lib.tbk_calib_gather_pairs(0, foot_gb << 30, inf, waves, 1 << 28, 3, C.byref(lps)))
            row[f"waves{waves}_inflight{inf}"] = round(lps.value / 1e9, 2)
    old = C.c_double(); ms = C.c_double()
    check(lib.tbk_calib_gather(0, foot_gb << 30, 32, 2, 2, 1 << 28, 3, C.byref(old), C.byref(ms)))
    row["round4_shape_256_thread_blocks_inflight2"] = round(old.value / 1e9, 2)
    row["best"] = max(v for k, v in row.items() if k.startswith("waves"))
    res["random_lines_Glines_per_s"][f"{foot_gb}GB"] = row
for unroll in (1, 2, 4, 8):
    for blocks in (1024, 2048, 4096, 8192, 16384):
        bps = C.c_double()
        check(lib.tbk_calib_stream_nt(0, 16 << 30, unroll, blocks, 4, C.byref(bps)))
        res["stream_GBps"][f"unroll{unroll}_blocks{blocks}"] = round(bps.value / 1e9, 1)
bps = C.c_double()
check(lib.tbk_calib_stream(0, 16 << 30, 4, C.byref(bps)))
res["stream_GBps"]["round4_kernel"] = round(bps.value / 1e9, 1)
res["stream_GBps"]["best"] = max(v for k, v in res["stream_GBps"].items() if k.startswith("unroll"))
print(json.dumps(res))
