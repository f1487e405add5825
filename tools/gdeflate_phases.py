import ctypes as C, os, sys, json
sys.path.insert(0, "/root/repo")
os.environ["TBK_LIBRARY"] = "/root/repo/trio_binning_amd/csrc/dbg/libtbk_phases.so"
sys.argv = ["x", "--reps", "3"] + sys.argv[1:]
exec(open("/root/repo/tools/measure_gdeflate.py").read())
from trio_binning_amd._lib import lib
out = (C.c_ulonglong * 16)()
lib.tbk_gdeflate_phases_(out)
names = ["stage+clear", "same", "histogram", "tree", "codes", "header", "count+scan", "codes out", "-"]
tot = sum(out[:9])
print({n: round(100.0 * out[i] / tot, 1) for i, n in enumerate(names)}, "cycles per block:", tot / (3 * 9000))
