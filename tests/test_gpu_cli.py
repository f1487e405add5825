"""End-to-end CLI on the reference's toy inputs (BASELINE.json configs[0] inputs, run on
the GPU path): stdout TSV and decompressed bins must equal what the real reference
produced (tests/golden/toy_cli.json, diff_vectors.json)."""
import gzip
import hashlib
import os
from unittest.mock import patch

import pytest

from conftest import DATA, load_golden

pytestmark = pytest.mark.gpu


def _run(argv, capsys):
    from trio_binning_amd.classify_by_kmers import main

    with patch("sys.argv", ["classify-by-kmers"] + argv):
        main()
    out, _ = capsys.readouterr()
    return out


@pytest.mark.parametrize("reads_name", ["test.ccs.fastq.gz", "test.fa", "test.fastq"])
def test_toy_cli_gzip_mode(gpu, capsys, tmp_path, reads_name):
    g = load_golden("toy_cli.json")[reads_name]
    pa, pb, pu = (str(tmp_path / x) for x in ("hapA", "hapB", "unclassified"))
    out = _run([os.path.join(DATA, reads_name), os.path.join(DATA, "hapA.txt"), os.path.join(DATA, "hapB.txt"),
                "--haplotype-a-out-prefix", pa, "--haplotype-b-out-prefix", pb, "--unclassified-out-prefix", pu], capsys)
    assert out == g["stdout"]
    assert sorted(os.listdir(tmp_path)) == sorted(g["files"])
    for fn, meta in g["files"].items():
        body = gzip.open(os.path.join(tmp_path, fn), "rb").read()
        assert len(body) == meta["size"], fn
        assert hashlib.sha256(body).hexdigest() == meta["sha256"], fn
        if "text" in meta:
            assert body.decode() == meta["text"]


def test_reference_cli_test_shape(gpu, capsys, tmp_path):
    """The reference's own CLI test (tests/test_classify_by_kmers.py:19-57) with the
    --no-gzip-output defect fixed: the hapA file holds exactly the A read."""
    from trio_binning_amd.seq import readfq

    pa, pb, pu = (str(tmp_path / x) for x in ("hapA", "hapB", "hapU"))
    out = _run([os.path.join(DATA, "test.ccs.fastq.gz"), os.path.join(DATA, "hapA.txt"), os.path.join(DATA, "hapB.txt"),
                "--haplotype-a-out-prefix", pa, "--haplotype-b-out-prefix", pb, "--unclassified-out-prefix", pu,
                "--no-gzip-output"], capsys)
    assert out == load_golden("toy_cli.json")["test.ccs.fastq.gz"]["stdout"]
    names = {fn: [r.name for r in readfq(open(os.path.join(tmp_path, fn)))] for fn in sorted(os.listdir(tmp_path))}
    assert names == {"hapA.fastq": ["m64234e_220609_193909/2/ccs"], "hapB.fastq": ["m64234e_220609_193909/3/ccs"],
                     "hapU.fastq": ["m64234e_220609_193909/6/ccs"]}
    # each bin equals the corresponding gzip-mode bin of the reference
    g = load_golden("toy_cli.json")["test.ccs.fastq.gz"]["files"]
    for ours, theirs in (("hapA.fastq", "hapA.fastq.gz"), ("hapB.fastq", "hapB.fastq.gz"), ("hapU.fastq", "unclassified.fastq.gz")):
        assert hashlib.sha256(open(os.path.join(tmp_path, ours), "rb").read()).hexdigest() == g[theirs]["sha256"]


@pytest.mark.parametrize("k", [21, 32])
def test_differential_cli(gpu, capsys, tmp_path, k):
    """150 reads with planted hits through the CLI: TSV text and the three bins equal the
    real reference's (recorded in diff_vectors.json)."""
    v = next(x for x in load_golden("diff_vectors.json") if x["k"] == k)
    fa, fb, fq = tmp_path / "la.txt", tmp_path / "lb.txt", tmp_path / f"reads{k}.fa"
    fa.write_text("".join(x + "\n" for x in v["list_a"]))
    fb.write_text("".join(x + "\n" for x in v["list_b"]))
    with open(fq, "w") as fh:
        for i, s in enumerate(v["reads"]):
            fh.write(f">r{i} some comment\n{s}\n")
    od = tmp_path / "out"
    od.mkdir()
    out = _run([str(fq), str(fa), str(fb), "--haplotype-a-out-prefix", str(od / "hapA"),
                "--haplotype-b-out-prefix", str(od / "hapB"), "--unclassified-out-prefix", str(od / "unclassified")], capsys)
    assert out == v["cli_stdout"]
    assert sorted(os.listdir(od)) == sorted(v["cli_bins"])
    for fn, digest in v["cli_bins"].items():
        assert hashlib.sha256(gzip.open(od / fn, "rb").read()).hexdigest() == digest, fn


@pytest.mark.parametrize("k,wide", [(21, 0), (21, 1), (32, 1), (21, 2)])
def test_differential_cli_through_the_entry_layouts(gpu, capsys, tmp_path, monkeypatch, k, wide):
    """The same recorded reference output with the table in entry layout (narrow entries, and the 16-byte ones k = 32
    needs) and in short keys (wide = 2): the native loop, its packed batches and short reads' multi-read passes over
    tbk_probe_entry_kernel."""
    import trio_binning_amd.classify_by_kmers as cbk

    monkeypatch.setenv("TBK_SHORT" if wide == 2 else "TBK_ENTRY", "1")
    if wide == 1:
        monkeypatch.setenv("TBK_ENTRY_WIDE", "1")
    monkeypatch.setattr(cbk, "_BATCH_BASES", 2000)
    monkeypatch.setattr(cbk, "_BATCH_READS", 40)
    test_differential_cli(gpu, capsys, tmp_path, k)


def test_stats_line_goes_to_stderr_only(gpu, capsys, tmp_path, monkeypatch):
    import json

    monkeypatch.setenv("TBK_STATS", "1")
    g = load_golden("toy_cli.json")["test.fastq"]
    from trio_binning_amd.classify_by_kmers import main

    with patch("sys.argv", ["classify-by-kmers", os.path.join(DATA, "test.fastq"), os.path.join(DATA, "hapA.txt"),
                            os.path.join(DATA, "hapB.txt"), "--haplotype-a-out-prefix", str(tmp_path / "a"),
                            "--haplotype-b-out-prefix", str(tmp_path / "b"), "--unclassified-out-prefix", str(tmp_path / "u")]):
        main()
    out, err = capsys.readouterr()
    assert out == g["stdout"]
    line = [l for l in err.splitlines() if l.startswith("tbk-stats ")][-1]
    st = json.loads(line[len("tbk-stats "):])
    assert st["reads"] == 4 and st["bases"] == 220 and st["batches"] == 1


def test_small_batches_keep_input_order(gpu, capsys, tmp_path, monkeypatch):
    """Force many tiny batches through the streaming ring: outputs identical."""
    import trio_binning_amd.classify_by_kmers as cbk

    monkeypatch.setattr(cbk, "_BATCH_BASES", 300)
    monkeypatch.setattr(cbk, "_BATCH_READS", 7)
    test_differential_cli(gpu, capsys, tmp_path, 21)


def test_synthetic_prefix_through_the_file_path(gpu, capsys, tmp_path):
    """a prefix of the synthetic bench workload goes through files — text k-mer lists,
    a FASTQ of GPU-generated reads — and the CLI must report exactly the counts that the
    device-resident path (bench.py's path) gets on the same bytes."""
    import ctypes as C

    import numpy as np

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    dev, k, n_list, R, L = 0, 21, 20000, 256, 3000

    def dalloc(n):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, n, C.byref(p)))
        return p.value

    d_keys = dalloc(2 * n_list * 8)
    check(lib.tbk_synth_keys_device(dev, 0x5EED0001, 0, 2 * n_list, k, C.c_void_p(d_keys)))
    h_keys = np.empty(2 * n_list, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(dev, h_keys.ctypes.data, C.c_void_p(d_keys), h_keys.nbytes))
    a = kmers.HashSet.from_device_keys(d_keys, n_list, k)
    b = kmers.HashSet.from_device_keys(d_keys + n_list * 8, n_list, k)
    total = R * L
    d_bases, d_offs, d_counts = dalloc(total + 32), dalloc((R + 1) * 8), dalloc(R * 8)
    check(lib.tbk_synth_reads_device(dev, 0x5EED0002, 0, R, L, 0x5EED0001, n_list, n_list, k, 30, 3, C.c_void_p(d_bases), C.c_void_p(d_offs)))
    with kmers.Classifier(a, b) as cls:
        cls.classify_device(d_bases, d_offs, R, total, d_counts)
        cls.sync()
    counts = np.zeros((R, 2), dtype=np.int32)
    check(lib.tbk_memcpy_d2h(dev, counts.ctypes.data, C.c_void_p(d_counts), counts.nbytes))
    bases = np.empty(total, dtype=np.uint8)
    check(lib.tbk_memcpy_d2h(dev, bases.ctypes.data, C.c_void_p(d_bases), total))
    for p in (d_keys, d_bases, d_offs, d_counts):
        check(lib.tbk_device_free(dev, C.c_void_p(p)))

    def decode(keys):
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        out = np.empty((keys.size, k + 1), dtype=np.uint8)
        for i in range(k):
            out[:, i] = lut[((keys >> np.uint64(2 * i)) & np.uint64(3)).astype(np.int64)]
        out[:, k] = 10
        return out.tobytes()

    (tmp_path / "hapA.txt").write_bytes(decode(h_keys[:n_list]))
    (tmp_path / "hapB.txt").write_bytes(decode(h_keys[n_list:]))
    with open(tmp_path / "reads.fastq", "wb") as fh:
        for r in range(R):
            fh.write(b"@s%d\n" % r + bases[r * L:(r + 1) * L].tobytes() + b"\n+\n" + b"I" * L + b"\n")
    od = tmp_path / "out"
    od.mkdir()
    out = _run([str(tmp_path / "reads.fastq"), str(tmp_path / "hapA.txt"), str(tmp_path / "hapB.txt"),
                "--haplotype-a-out-prefix", str(od / "hapA"), "--haplotype-b-out-prefix", str(od / "hapB"),
                "--unclassified-out-prefix", str(od / "unc")], capsys)
    lines = out.splitlines()
    assert len(lines) == R
    sa, sb, bins = kmers.score_and_bin(counts, n_list, n_list)
    for r, line in enumerate(lines):
        name, bin_, x, y = line.split("\t")
        assert (name, bin_, float(x), float(y)) == (f"s{r}", chr(bins[r]), float(sa[r]), float(sb[r]))
    n_in_bins = sum(len(list(readfq_names(od / f))) for f in ("hapA.fastq.gz", "hapB.fastq.gz", "unc.fastq.gz"))
    assert n_in_bins == R and counts.sum() > 20 * R


def readfq_names(path):
    import gzip

    from trio_binning_amd.seq import readfq

    with gzip.open(path, "rt") as fh:
        for rec in readfq(fh):
            yield rec.name
