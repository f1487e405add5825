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


# ---- native batch I/O (C-ABI tbk_fastx_* / tbk_bin_writer_*) --------------------------------
# The same records and the same output bytes as readfq / Read.print above, produced by the
# library in the batch layout the classifier consumes; used by the CLI driver.  Imported
# lazily so that this module stays importable without the built library.
class Batch:
    """One batch of records read by ``BatchReader`` (owned by the native library)."""

    def __init__(self):
        import ctypes as C

        from ._lib import check, lib

        h = C.c_void_p()
        check(lib.tbk_fastx_batch_create(C.byref(h)))
        self._h = h
        self.n_reads = 0

    def _view(self):
        import ctypes as C

        from ._lib import check, lib

        n = C.c_uint64()
        ptrs = [C.c_void_p() for _ in range(7)]
        check(lib.tbk_fastx_batch_view(self._h, C.byref(n), *[C.byref(p) for p in ptrs]))
        return n.value, [p.value for p in ptrs]

    def arrays(self):
        """(bases, base_off, names, name_off, quals, qual_off, has_qual) as numpy views that
        stay valid until the batch is refilled or destroyed.  A borrowed batch has no sequence / quality
        arrays of its own: `bases` and `quals` are then copies gathered from the reader's mapping."""
        import ctypes as C

        import numpy as np

        from ._lib import check, lib

        n, (bases, boff, names, noff, quals, qoff, hq) = self._view()

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            ct = {np.uint8: C.c_uint8, np.uint64: C.c_uint64}[dtype]
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(count,))

        base_off = arr(boff, n + 1, np.uint64)
        name_off = arr(noff, n + 1, np.uint64)
        qual_off = arr(qoff, n + 1, np.uint64)
        if self.borrowed:
            bases_a = np.zeros(int(base_off[-1]), dtype=np.uint8)
            quals_a = np.zeros(int(qual_off[-1]), dtype=np.uint8)
            check(lib.tbk_fastx_batch_gather(self._h, bases_a.ctypes.data, bases_a.size, quals_a.ctypes.data, quals_a.size))
        else:
            bases_a, quals_a = arr(bases, int(base_off[-1]), np.uint8), arr(quals, int(qual_off[-1]), np.uint8)
        return (bases_a, base_off, arr(names, int(name_off[-1]), np.uint8), name_off, quals_a, qual_off, arr(hq, n, np.uint8))

    @property
    def borrowed(self) -> bool:
        """The batch's records lie in its reader's mapping of the input (``BatchReader(..., borrowing=True)``)."""
        from ._lib import lib

        return bool(lib.tbk_fastx_batch_borrowed(self._h))

    def pointers(self):
        """(bases_ptr, base_off_ptr) for tbk_stream_submit."""
        if self.borrowed:
            raise ValueError("a borrowed batch has no ASCII bases array: submit its packed form (packed_pointers)")
        _, p = self._view()
        return p[0], p[1]

    def packed_pointers(self):
        """(codes_ptr, exc_chunk_ptr, exc_mask_ptr, n_exc) of the batch's packed transfer form, or None when the
        reader did not make one (``BatchReader(..., packing=True)``)."""
        import ctypes as C

        from ._lib import check, lib

        codes, ec, em, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        check(lib.tbk_fastx_batch_packed(self._h, C.byref(codes), C.byref(ec), C.byref(em), C.byref(n)))
        if not codes.value:
            return None
        return codes.value, ec.value, em.value, n.value

    def packed_arrays(self):
        """(codes, exc_chunk, exc_mask) as numpy copies (tests), or None."""
        import ctypes as C

        import numpy as np

        p = self.packed_pointers()
        if p is None:
            return None
        total = int(self.arrays()[1][-1])
        n_chunks = (total + 15) // 16
        codes = np.ctypeslib.as_array(C.cast(p[0], C.POINTER(C.c_uint32)), (max(n_chunks, 1),))[:n_chunks].copy()
        ec = np.ctypeslib.as_array(C.cast(p[1], C.POINTER(C.c_uint32)), (max(p[3], 1),))[:p[3]].copy()
        em = np.ctypeslib.as_array(C.cast(p[2], C.POINTER(C.c_uint16)), (max(p[3], 1),))[:p[3]].copy()
        return codes, ec, em

    def reads(self) -> List[Read]:
        """The batch as ``Read`` objects (tests; the CLI never materialises them)."""
        bases, boff, names, noff, quals, qoff, hq = self.arrays()
        out = []
        for i in range(self.n_reads):
            name = bytes(names[int(noff[i]):int(noff[i + 1])]).decode()
            sq = bytes(bases[int(boff[i]):int(boff[i + 1])]).decode()
            ql = bytes(quals[int(qoff[i]):int(qoff[i + 1])]).decode() if hq[i] else None
            out.append(Read(name, sq, ql))
        return out

    def close(self):
        if self._h is not None and self._h.value:
            from ._lib import lib

            lib.tbk_fastx_batch_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchReader:
    """Native FASTA/FASTQ(.gz) reader: ``next_batch(batch, max_bases, max_reads)`` fills a
    ``Batch`` and returns the number of records (0 at end of input)."""

    def __init__(self, filename: str, packing: bool = False, borrowing: bool = False, device: Optional[int] = None):
        """``device``: BGZF input is inflated on that GPU (``tbk_fastx_set_device``: what ``classify-by-kmers`` does by itself);
        None: on the host's threads.  ``inflates_on_device`` says which it is."""
        import ctypes as C
        import os

        from ._lib import check, lib

        h = C.c_void_p()
        check(lib.tbk_fastx_open(os.fsencode(filename), C.byref(h)))
        self._h = h
        if device is not None:
            check(lib.tbk_fastx_set_device(h, device))
        self.inflates_on_device = bool(lib.tbk_fastx_inflates_on_device(h)) if hasattr(lib, "tbk_fastx_inflates_on_device") else False
        if packing:  # batches also carry the packed transfer form of their bases (Batch.packed_pointers)
            check(lib.tbk_fastx_set_packing(h, 1))
        if borrowing:  # batches of a plain FASTQ file leave their records in the reader's mapping (what the native loop does):
            check(lib.tbk_fastx_set_borrowing(h, 1))  # no sequence / quality arrays; valid until close()

    def next_batch(self, batch: Batch, max_bases: int = 0, max_reads: int = 0) -> int:
        import ctypes as C

        from ._lib import check, lib

        check(lib.tbk_fastx_next(self._h, batch._h, max_bases, max_reads))
        n = C.c_uint64()
        check(lib.tbk_fastx_batch_view(batch._h, C.byref(n), None, None, None, None, None, None, None))
        batch.n_reads = n.value
        return n.value

    def close(self):
        if self._h is not None and self._h.value:
            from ._lib import lib

            lib.tbk_fastx_close(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BinWriter:
    """Native writer of the three bins (same names, same decompressed bytes as
    ``open_outfiles`` + ``Read.print``); gzip members are deflated in parallel."""

    def __init__(self, haplotype_a_prefix: str, haplotype_b_prefix: str, unclassified_prefix: str,
                 outfile_extension: str, gzip_output: bool, level: int = -1, threads: int = 0, device: Optional[int] = None):
        """``device``: code the gzip members on that GPU (``tbk_bin_writer_use_device``: what ``classify-by-kmers`` does by itself);
        None: on the host's threads.  ``gpu_encoder`` says which it is."""
        import ctypes as C
        import os

        from ._lib import check, lib

        self.names = output_names(haplotype_a_prefix, haplotype_b_prefix, unclassified_prefix, outfile_extension, gzip_output)
        h = C.c_void_p()
        check(lib.tbk_bin_writer_open(*[os.fsencode(n) for n in self.names], int(gzip_output), level, threads, C.byref(h)))
        self._h = h
        if device is not None:
            check(lib.tbk_bin_writer_use_device(h, device))
        self.gpu_encoder = bool(lib.tbk_bin_writer_encoder(h)) if hasattr(lib, "tbk_bin_writer_encoder") else False

    def write(self, batch: Batch, bins: bytes) -> None:
        from ._lib import check, lib

        check(lib.tbk_bin_writer_write(self._h, batch._h, bins))

    def close(self) -> None:
        if self._h is not None and self._h.value:
            from ._lib import check, lib

            h, self._h = self._h, None
            check(lib.tbk_bin_writer_close(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_tsv(batch: Batch, bins: bytes, score_a, score_b) -> str:
    """``name\\tbin\\tscoreA\\tscoreB`` lines for a batch, floats formatted as Python's str()."""
    import ctypes as C

    from ._lib import check, lib

    need = C.c_size_t()
    check(lib.tbk_format_tsv(batch._h, bins, score_a.ctypes.data, score_b.ctypes.data, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value + 1)
    check(lib.tbk_format_tsv(batch._h, bins, score_a.ctypes.data, score_b.ctypes.data, buf, need.value + 1, C.byref(need)))
    return buf.raw[: need.value].decode()


def gzip_members_device(pieces, device: int = 0):
    """Each of ``pieces`` (bytes) as one gzip member, coded on the GPU (``tbk_gzip_members_device``: the bin writer's encoder by
    itself).  Returns the members as a list of bytes; ``gzip.decompress`` of each gives the piece back."""
    import ctypes as C

    from ._lib import check, lib

    text = b"".join(pieces)
    n = len(pieces)
    lens = (C.c_uint64 * max(1, n))(*[len(p) for p in pieces])
    out_lens = (C.c_uint64 * max(1, n))()
    need = C.c_uint64()
    cap = len(text) + len(text) // 8 + 1100 * (len(text) // 8192 + 1) + 64 * n + 1024
    buf = C.create_string_buffer(cap)
    check(lib.tbk_gzip_members_device(device, text, lens, n, buf, cap, out_lens, C.byref(need)))
    out, at = [], 0
    for i in range(n):
        out.append(C.string_at(C.addressof(buf) + at, out_lens[i]))
        at += out_lens[i]
    return out


def bgzf_inflate_device(data: bytes, device: int = 0) -> bytes:
    """The text of a bgzf file (bytes), inflated on the GPU (``tbk_bgzf_inflate_device``: the reader's bgzf path by itself)."""
    import ctypes as C

    from ._lib import check, lib

    n = C.c_uint64()
    status = lib.tbk_bgzf_inflate_device(device, data, len(data), None, 0, C.byref(n))
    if n.value == 0:
        check(status)
        return b""
    buf = C.create_string_buffer(n.value)
    check(lib.tbk_bgzf_inflate_device(device, data, len(data), buf, n.value, C.byref(n)))
    return C.string_at(C.addressof(buf), n.value)
