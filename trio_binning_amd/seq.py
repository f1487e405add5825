"""FASTA/FASTQ(.gz) records in, three bins out.

Host-side mirror of the reference's ``trio_binning.seq`` (src/trio_binning/seq.py):
``Read``, ``readfq``, ``open_fastx_read`` and ``open_outfiles`` with the same record
semantics and output bytes, written as an explicit line state machine.  The record rules
below are the observable behaviour of the reference parser (seq.py:45-83), including its
corner cases, because the bins must be byte-identical:

* a record starts at a line whose first character is ``>`` or ``@``; the name is the
  header without that character, cut at the first space (not tab);
* every line loses its last character, newline or not (a file without a final newline
  loses its last base / quality value);
* sequence lines run until a line starting with ``@``, ``+`` or ``>``; only ``+`` makes the
  record FASTQ;
* quality lines are taken until their summed length reaches the sequence length;
  running out of input first turns the record into FASTA (qual ``None``) and ends parsing;
* a header or separator line that is empty after losing its last character counts as
  "no line" (end of input).
"""
import gzip
import sys
from dataclasses import dataclass
from typing import IO, Iterable, Iterator, List, Optional, TextIO, Tuple, Union


@dataclass
class Read:
    """A fastx read"""

    name: str
    """The name of the read"""
    seq: str
    """The sequence of the read"""
    qual: Optional[str] = None
    """The quality score string of the read"""

    def __str__(self) -> str:
        # reference seq.py:27-31: FASTQ when there is a non-empty quality string
        if self.qual:
            return "@" + self.name + "\n" + self.seq + "\n+\n" + self.qual
        return ">" + self.name + "\n" + self.seq

    def print(self, file: TextIO = sys.stdout) -> None:
        """Print the read in fastq format if it has qualities, else fasta (seq.py:33-42)."""
        file.write(str(self))
        file.write("\n")


_SEEK, _SEQ, _QUAL = 0, 1, 2


def readfq(fp: Iterable[str]) -> Iterator[Read]:
    """Read a fast[aq] stream, yielding a ``Read`` per record (reference seq.py:45-83)."""
    state = _SEEK
    name = ""
    parts: List[str] = []
    seq = ""
    have = 0
    for line in fp:
        head = line[0] if line else ""
        body = line[:-1]
        if state == _SEEK:
            if head == ">" or head == "@":
                if not body:  # header that vanishes with its last character: input ends here
                    return
                name = body[1:].partition(" ")[0]
                parts = []
                state = _SEQ
            continue
        if state == _SEQ:
            if head == "@" or head == "+" or head == ">":
                seq = "".join(parts)
                if not body:  # separator/header reduced to nothing: FASTA record, then stop
                    yield Read(name, seq, None)
                    return
                if head == "+":
                    parts = []
                    have = 0
                    state = _QUAL
                else:
                    yield Read(name, seq, None)
                    name = body[1:].partition(" ")[0]
                    parts = []
            else:
                parts.append(body)
            continue
        # _QUAL
        parts.append(body)
        have += len(line) - 1
        if have >= len(seq):
            yield Read(name, seq, "".join(parts))
            state = _SEEK
    # input exhausted
    if state == _SEQ:
        yield Read(name, "".join(parts), None)
    elif state == _QUAL:
        yield Read(name, seq, None)  # not enough quality: emitted as FASTA (seq.py:81-83)


def open_fastx_read(filename: str) -> Iterator[Read]:
    """Open a fasta/q(.gz) file for reading (reference seq.py:86-92): gzip by file name,
    text mode with universal newlines."""
    if filename.endswith(".gz"):
        return readfq(gzip.open(filename, "rt"))
    return readfq(open(filename, "r"))


TextOrGzip = Union[TextIO, IO[str]]


def output_names(
    haplotype_a_prefix: str,
    haplotype_b_prefix: str,
    unclassified_prefix: str,
    outfile_extension: str,
    gzip_output: bool,
) -> Tuple[str, str, str]:
    """File names of the three bins: prefix + extension (+ ``.gz``), seq.py:117-134."""
    tail = outfile_extension + (".gz" if gzip_output else "")
    return haplotype_a_prefix + tail, haplotype_b_prefix + tail, unclassified_prefix + tail


def open_outfiles(
    haplotype_a_prefix: str,
    haplotype_b_prefix: str,
    unclassified_prefix: str,
    outfile_extension: str,
    gzip_output: bool,
) -> Tuple[TextOrGzip, TextOrGzip, TextOrGzip]:
    """Open the three output bins (reference seq.py:98-136).

    Deviation, on purpose: with ``gzip_output=False`` the reference opens its haplotype-B
    handle on the haplotype-A file name (seq.py:129), so B reads overwrite the front of
    the A file and no B file is created.  Here B goes to the B file.  Gzip mode (the
    default) is identical to the reference: ``gzip.open(name + ".gz", "wt")``.
    """
    names = output_names(haplotype_a_prefix, haplotype_b_prefix, unclassified_prefix, outfile_extension, gzip_output)
    if gzip_output:
        return tuple(gzip.open(n, "wt") for n in names)  # type: ignore[return-value]
    return tuple(open(n, "w") for n in names)  # type: ignore[return-value]
