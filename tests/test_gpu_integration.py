"""INTEGRATION.md's binding snippets, executed verbatim against libtbk_hip.so on the GPU.

Section B (the reference's kmers.py patched to the tbk_* entry points) and section C (the reference's
own symbol names and struct layout, include/kmers_compat.h) each hold one python block tagged
`# integration-snippet: <letter>`.  The blocks are this repository's text - what a maintainer of the
reference would write - and the checks are the reference's known answers (tests/test_kmers.py:8-53)."""
import os
import re

import pytest

from conftest import DATA, ROOT, load_golden

pytestmark = pytest.mark.gpu

READ = "CTTATCATGTCTTTGTTTTCAAAGCTTCTTAGAGGTTTTTTTTTTTGGTGTTAATTGGCATAAATTATGGCT"


def snippet(letter):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(# integration-snippet: %s\n.*?)```" % letter, text, flags=re.S)
    assert len(blocks) == 1, f"INTEGRATION.md must hold exactly one block tagged {letter}"
    return blocks[0]


@pytest.mark.parametrize("letter", ["B", "C"])
def test_integration_snippet_runs_the_reference_kats(gpu, letter, capfd, tmp_path):
    ns = {"__file__": os.path.join(ROOT, "trio_binning_amd", "kmers.py"), "__name__": "reference_kmers_patched"}
    exec(compile(snippet(letter), f"INTEGRATION.md[{letter}]", "exec"), ns)
    a = ns["create_kmer_hash_set"](os.path.join(DATA, "hapA.txt"))
    b = ns["create_kmer_hash_set"](os.path.join(DATA, "hapB.txt"))
    assert ns["get_number_kmers_in_set"](a) == 4 and ns["get_number_kmers_in_set"](b) == 3
    assert ns["count_kmers_in_read"](READ, a, b) == (2, 1)
    kat = load_golden("kat.json")
    for case in kat["count_kmers_in_read"]:
        assert list(ns["count_kmers_in_read"](case["read"], a, b)) == case["counts"]
    if letter == "C":
        # kmers.py:41-59's struct mirror reads this library's struct
        assert (a.contents.k, a.contents.num_kmers, a.contents.hash_size) == (21, 4, 5)
        assert (b.contents.k, b.contents.num_kmers, b.contents.hash_size) == (21, 3, 4)
        assert not a.contents.kmers and not a.contents.full  # the keys live in HBM
        for kmer, value in kat["kmer_to_int"]:
            assert ns["kmer_to_int"](kmer) == value
        for kmer, value in kat["reverse_complement"]:
            # (the reference's binding, kmers.py:89-93, lets the C function write into a `bytes` object; for a k-mer of ONE base that
            # object is CPython's shared b"x", and every b"x" of the process reads "G" from then on - found when a later test compared one.
            # The one-base vector is the oracle's business, tests/test_oracle_golden.py; here it would poison the interpreter.)
            if len(kmer) == 1:
                continue
            assert ns["reverse_complement"](kmer) == value
        with pytest.raises(IOError):
            ns["create_kmer_hash_set"](os.path.join(DATA, "no_such_list.txt"))
        # a failure inside the library: NULL handle, counts left at -1 (the C signatures have no error channel)
        bad = tmp_path / "k40.txt"                              # a file, but its first line gives k = 40 > 32
        bad.write_text("A" * 40 + "\n")
        h = ns["create_kmer_hash_set"](str(bad))
        assert not h
        with pytest.raises(ValueError):
            h.contents
    else:
        with pytest.raises(IOError):
            ns["create_kmer_hash_set"](os.path.join(DATA, "no_such_list.txt"))
    capfd.readouterr()
