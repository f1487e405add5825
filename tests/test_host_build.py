"""Build-time properties of the probe kernels, read from the built object's metadata (no GPU needed).

The single-read probe kernel - the dominant kernel - must fit five waves per SIMD (<= 96 VGPRs) without a
single spilled VGPR: a probe kernel that touches scratch memory at all ran 18-22 ms from one stream to the next
where the spill-free one runs 17.1-17.3 (EXPERIMENTS.md, round 3); the whole-line variants up to W = 6 too.  The multi-read kernel must not spill VGPRs either
(<= 128, four waves); the two-read kernels of the front layout (up to W = 7) fit five as well."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

OBJ = os.path.join(ROOT, "trio_binning_amd", "csrc", "build", "tbk_kernels.o")
TOOL = os.path.join(ROOT, "tools", "kernel_regs.sh")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="needs the ROCm LLVM tools")
def test_probe_kernels_do_not_spill_vector_registers(built):
    assert os.path.isfile(OBJ), "the library was built from objects under csrc/build"
    out = subprocess.run(["bash", TOOL, OBJ, "tbk_probe_kernel"], capture_output=True, text=True, check=True).stdout
    rows = []
    for line in out.splitlines():
        m = re.match(r"(\S+) vgpr (\d+) sgpr_spill (\d+) vgpr_spill (\d+) lds (\d+)", line)
        if m:
            name = m.group(1)
            flags = re.search(r"tbk_probe_kernelILi(\d)E((?:Lb[01]E){4,5})", name)
            assert flags, name
            bits = [c == "1" for c in re.findall(r"Lb([01])E", flags.group(2))]
            rows.append({"name": name, "w": int(flags.group(1)), "m64": bits[0], "samp": bits[1], "front": bits[2], "multi": bits[3], "two": len(bits) > 4 and bits[4],
                         "vgpr": int(m.group(2)), "vgpr_spill": int(m.group(4)), "lds": int(m.group(5))})
    assert len(rows) >= 60, len(rows)   # every (W, m-mer width, sampling rule, layout) variant, single- and multi-read
    for r in rows:
        assert r["vgpr_spill"] == 0, r                         # no scratch memory in any probe kernel
        five = not r["multi"] and ((r["front"] and r["w"] <= 7) if r["two"] else (r["front"] or 2 <= r["w"] <= 6))
        if five:
            assert r["vgpr"] <= 96, r                          # five waves per SIMD (whole lines: up to W = 6; two-read passes: front layout up to W = 7)
        else:
            assert r["vgpr"] <= 128, r                         # four
        assert r["lds"] <= 8192, r                             # 20 one-wave blocks per CU fit the 160 KB of LDS
