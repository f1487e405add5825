"""Host logic without a GPU: the FASTX reader/writer mirror against golden vectors taken
from the reference parser (tests/golden/readfq_vectors.json), plus the cases the
reference's own tests/test_seq.py holds."""
import gzip
import io
import os

from conftest import DATA, load_golden

from trio_binning_amd import seq


def test_read_fasta():
    # reference tests/test_seq.py:7-14
    n = 0
    for read in seq.readfq(open(os.path.join(DATA, "test.fa"))):
        n += 1
        assert read.name.startswith("read")
        assert read.seq.startswith("G")
        assert read.seq.endswith("A")
        assert not read.qual
    assert n == 3


def test_read_fastq():
    # reference tests/test_seq.py:17-24
    n = 0
    for read in seq.readfq(open(os.path.join(DATA, "test.fastq"))):
        n += 1
        assert read.name.startswith("read")
        assert read.seq.startswith("G")
        assert read.seq.endswith("A")
        assert read.qual.startswith("!")
    assert n == 4


def test_write_fasta():
    # reference tests/test_seq.py:27-32
    sio = io.StringIO()
    seq.Read("read1", "AGATAGAGGACTGA").print(file=sio)
    seq.Read("read2", "AGGGGATTTTATTA").print(file=sio)
    assert sio.getvalue() == ">read1\nAGATAGAGGACTGA\n>read2\nAGGGGATTTTATTA\n"


def test_write_fastq():
    # reference tests/test_seq.py:35-43
    sio = io.StringIO()
    seq.Read("read1", "AGATAGAGGACTGA", "%()%%%(%(++***").print(file=sio)
    seq.Read("read2", "AGGGGATTTTATTA", "++(*))*+%%%))(").print(file=sio)
    assert sio.getvalue() == (
        "@read1\nAGATAGAGGACTGA\n+\n%()%%%(%(++***\n"
        "@read2\nAGGGGATTTTATTA\n+\n++(*))*+%%%))(\n"
    )


def test_readfq_quirks_match_reference():
    g = load_golden("readfq_vectors.json")
    for name, case in g.items():
        if name == "crlf_file":
            continue
        recs = [[r.name, r.seq, r.qual] for r in seq.readfq(io.StringIO(case["text"]))]
        assert recs == case["records"], name
        out = io.StringIO()
        for r in seq.readfq(io.StringIO(case["text"])):
            r.print(file=out)
        assert out.getvalue() == case["printed"], name


def test_crlf_file(tmp_path):
    case = load_golden("readfq_vectors.json")["crlf_file"]
    p = tmp_path / "crlf.fa"
    p.write_bytes(bytes.fromhex(case["bytes_hex"]))
    assert [[r.name, r.seq, r.qual] for r in seq.open_fastx_read(str(p))] == case["records"]


def test_gz_and_plain_read_the_same(tmp_path):
    text = open(os.path.join(DATA, "test.fastq")).read()
    gz = tmp_path / "x.fastq.gz"
    with gzip.open(gz, "wt") as fh:
        fh.write(text)
    a = list(seq.open_fastx_read(os.path.join(DATA, "test.fastq")))
    b = list(seq.open_fastx_read(str(gz)))
    assert a == b and len(a) == 4


def test_ccs_reads_lengths():
    lens = [len(r.seq) for r in seq.open_fastx_read(os.path.join(DATA, "test.ccs.fastq.gz"))]
    assert lens == [20288, 9808, 14017]


def test_open_outfiles_names_and_modes(tmp_path):
    a, b, u = (str(tmp_path / x) for x in ("hapA", "hapB", "unc"))
    assert seq.output_names(a, b, u, ".fq", True) == (a + ".fq.gz", b + ".fq.gz", u + ".fq.gz")
    assert seq.output_names(a, b, u, "", False) == (a, b, u)
    outs = seq.open_outfiles(a, b, u, ".fa", True)
    for i, fh in enumerate(outs):
        seq.Read(f"r{i}", "ACGT").print(file=fh)
        fh.close()
    for i, n in enumerate(seq.output_names(a, b, u, ".fa", True)):
        assert gzip.open(n, "rt").read() == f">r{i}\nACGT\n"
    # no-gzip mode: the reference writes B into the A file (seq.py:129); we do not
    outs = seq.open_outfiles(a, b, u, ".fa", False)
    for i, fh in enumerate(outs):
        seq.Read(f"r{i}", "ACGT").print(file=fh)
        fh.close()
    assert open(b + ".fa").read() == ">r1\nACGT\n" and open(a + ".fa").read() == ">r0\nACGT\n"
