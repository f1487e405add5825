"""More of tests/test_gpu_inflate.py's fuzz than the suite runs (120 seeds of 40 mixed members each) and six 40 MB cases: run by hand on a GPU box."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_inflate as t
ok = 0
t0 = time.time()
skipped = 0
_bgzf = t.bgzf
def bgzf_that_fits(text, level=6, block=60000, *a, **kw):   # (noise under fixed Huffman codes grows by an eighth: 56 000 bytes still fit a BGZF member)
    return _bgzf(text, level, min(block, 56000), *a, **kw)
t.bgzf = bgzf_that_fits
for seed in range(10, 130):
    t.test_fuzz_against_zlib(None, seed)
    ok += 1
print("fuzz seeds passed:", ok, "in %.1f s" % (time.time() - t0))
# bigger windows: 40 MB of text per case, mixed levels/blocks
from trio_binning_amd import seq
import zlib
rng = np.random.default_rng(99)
for case in range(6):
    text = t.fastq(rng, 1400, 15000, ["hifi", "const"][case % 2])
    data = t.bgzf(text, [1, 4, 6, 9, 6, 2][case], [65280, 60000, 30000, 65280, 1000, 50000][case])
    assert seq.bgzf_inflate_device(data) == text, case
print("large cases ok")
