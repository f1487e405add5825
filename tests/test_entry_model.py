"""CPU model of the entry layout (tests/native/entry_model.cpp over csrc/tbk_common.h): a sequential build with the
insert rule of tbk_entry_insert_kernel, checked against plain set membership of canonical k-mers - list keys through
every (tied position, orientation) form, windows of reads on both strands asked the way the probe kernel asks,
hapA-over-hapB priority (c/kmers.c:245-299).  No GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("entry_model") / "entry_model")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "entry_model.cpp")], check=True)
    return exe


@pytest.mark.parametrize("k,w,crowd", [(21, 6, 0), (21, 6, 1), (21, 5, 1), (21, 4, 0), (22, 6, 0), (22, 4, 1), (23, 6, 1), (23, 5, 0), (24, 5, 1), (25, 4, 0)])
def test_entry_model_equals_set_membership(model, k, w, crowd):
    # seed 1: 2w t-mer positions ranked per span, seed 2: 3w where t stays at 4 or more (tbk_mz_span3: what the library builds)
    for seed, span3 in ((1, 0), (2, 1)):
        r = subprocess.run([model, str(k), str(w), str(seed), str(crowd), "0", str(span3)], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "mismatches 0" in r.stdout and "no entry layout" not in r.stdout, r.stdout


@pytest.mark.parametrize("k,w,crowd", [(31, 8, 0), (31, 8, 1), (31, 6, 1), (32, 7, 1), (29, 8, 0), (27, 6, 1), (26, 7, 0), (21, 6, 1)])
def test_wide_entry_model_equals_set_membership(model, k, w, crowd):
    for seed in (1, 2):
        r = subprocess.run([model, str(k), str(w), str(seed), str(crowd), "1"], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert r.stdout.startswith("wide ") and "mismatches 0" in r.stdout and "no entry layout" not in r.stdout, r.stdout


def test_entry_geometry_limits(model):
    # k = 27 and beyond: the flanks no longer fit a narrow entry's 31 bits at any useful span - those get wide entries
    for k in (27, 31, 32):
        r = subprocess.run([model, str(k), "6", "1", "0"], capture_output=True, text=True)
        assert r.returncode == 0 and "no entry layout" in r.stdout, r.stdout


@pytest.fixture(scope="module")
def short_model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("short_model") / "short_model")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "short_model.cpp")], check=True)
    return exe


@pytest.mark.parametrize("k,w,n_buckets", [(21, 6, 0), (21, 6, 65536), (21, 6, 300000), (21, 5, 0), (19, 5, 70001), (22, 4, 0), (23, 6, 0)])
def test_short_key_model_equals_set_membership(short_model, k, w, n_buckets):
    """Short keys (tbk_common.h): a 32-bit word plus its bucket names a k-mer exactly - list keys through every form, near
    misses that share a line with them, windows of both strands, crowded lines that spill into the overflow table."""
    for seed, span3 in ((1, 0), (2, 1)):   # 2w and 3w t-mer positions per span (tbk_mz_span3)
        r = subprocess.run([short_model, str(k), str(w), str(seed), str(n_buckets), str(span3)], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert r.stdout.startswith("short ") and "mismatches 0" in r.stdout, r.stdout
        assert " 0 in the overflow table" not in r.stdout, r.stdout


def test_short_key_geometry_limits(short_model):
    # below the fewest buckets the word has no room for r; k = 31 has no room at any table size
    for args in (("21", "6", "1", "65535"), ("23", "6", "1", "1000000"), ("31", "6", "1", "1000000000")):
        r = subprocess.run([short_model, *args], capture_output=True, text=True)
        assert r.returncode == 0 and "no short keys" in r.stdout, r.stdout


@pytest.fixture(scope="module")
def full_model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("full_model") / "full_model")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "full_model.cpp")], check=True)
    return exe


@pytest.mark.parametrize("k,w,per_line", [(31, 8, 2.0), (31, 8, 6.0), (29, 8, 12.0), (27, 7, 4.0), (25, 6, 3.0), (23, 6, 10.0), (21, 6, 2.0)])
def test_full_key_model_equals_set_membership(full_model, k, w, per_line):
    """Full keys (tbk_common.h): sequential inserts into 16-slot lines with the summary in slot 3, keys sent on by
    tbk_next_bucket; tbk_full_lookup_one(..., as_window = 0 and 1) must both equal set membership - at crowded loads too (up to
    twelve keys per line asked for: lines fill and keys walk), list keys through every tied position, near misses that share
    a line with them, windows of both strands."""
    for seed in (1, 2):
        r = subprocess.run([full_model, str(k), str(w), str(seed), str(per_line)], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert r.stdout.startswith("full ") and "mismatches 0" in r.stdout and "as_window disagreements 0" in r.stdout, r.stdout
        if per_line >= 6:
            assert " 0 walks past a line" not in r.stdout, r.stdout
