"""Host code of the library (list parser, FASTX reader, bin writer, TSV formatter, error paths)
under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only: the sanitized build is loaded
through TBK_LIBRARY in a child interpreter with the ASan runtime preloaded, and the native-I/O,
packer, multi-device dealer and CLI host tests are re-run against it."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _asan_runtime():
    hits = glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


@pytest.mark.skipif(_asan_runtime() is None, reason="clang ASan runtime not found")
def test_host_code_under_asan_ubsan(built):
    csrc = os.path.join(ROOT, "trio_binning_amd", "csrc")
    subprocess.run(["make", "-s", "-j4", "-C", csrc, "asan"], check=True)  # (eleven host sources, four at a time: the 8-CPU container runs the suite beside it)
    env = dict(os.environ,
               LD_PRELOAD=_asan_runtime(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98",
               TBK_TEST_LIGHT="1",
               TBK_LIBRARY=os.path.join(csrc, "build_asan", "libtbk_hip_asan.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_host_native_io.py"),
                        os.path.join(ROOT, "tests", "test_host_pack.py"),
                        os.path.join(ROOT, "tests", "test_multi_cpu.py"),
                        os.path.join(ROOT, "tests", "test_host_cli.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
