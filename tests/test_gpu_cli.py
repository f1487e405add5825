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


def test_small_batches_keep_input_order(gpu, capsys, tmp_path, monkeypatch):
    """Force many tiny batches through the streaming ring: outputs identical."""
    import trio_binning_amd.classify_by_kmers as cbk

    monkeypatch.setattr(cbk, "_BATCH_BASES", 300)
    monkeypatch.setattr(cbk, "_BATCH_READS", 7)
    test_differential_cli(gpu, capsys, tmp_path, 21)
