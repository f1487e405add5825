"""Counting parental k-mers in reads on an MI355X.

Host-side mirror of the reference's ``trio_binning.kmers`` (src/trio_binning/kmers.py):
the same five functions with the same argument meaning and error behaviour, bound to the
HIP library's C-ABI (include/tbk.h) instead of the reference's c/kmers.c, plus the batch
interface the GPU needs (``Classifier``).

>>> hapA = kmers.create_kmer_hash_set("tests/data/hapA.txt")
>>> hapB = kmers.create_kmer_hash_set("tests/data/hapB.txt")
>>> kmers.count_kmers_in_read("CTTATCATGTCTTTGTTTTCAAAGCTTC...", hapA, hapB)
(2, 1)

Every function that computes needs a visible MI355X; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from os.path import isfile
from typing import Iterable, Optional, Sequence, Tuple


class _LazyNumpy:
    """numpy, imported at first use: the command line's path (lists from files, the native loop) needs none of it,
    and its import is a quarter of a second of a run that takes under three."""

    def __getattr__(self, name):
        import numpy

        globals()["np"] = numpy
        return getattr(numpy, name)


np = _LazyNumpy()

from . import _lib
from ._lib import check, lib


def default_device() -> int:
    """Device index used when none is given: TBK_DEVICE, else the first of TBK_DEVICES, else
    LOCAL_RANK, else 0."""
    for var in ("TBK_DEVICE", "TBK_DEVICES", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v not in (None, ""):
            return int(v.split(",")[0])
    return 0


class _Contents:
    """Stand-in for ``POINTER(_HashSet).contents`` (reference kmers.py:41-59,159)."""

    def __init__(self, hs):
        self._hs = hs

    @property
    def num_kmers(self) -> int:
        return self._hs.num_kmers

    @property
    def k(self) -> int:
        return self._hs.k


class HashSet:
    """A parental k-mer list resident in HBM (packed 64-bit keys).

    Plays the role of the reference's ``HashSet`` pointer type (kmers.py:96-101): what
    ``create_kmer_hash_set`` returns and ``count_kmers_in_read`` accepts.  Pairing two of
    them in a ``Classifier`` hashes them into the open-addressing tables the probe kernel
    reads; ``contains`` hashes this list on its own.
    """

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def from_file(cls, path: str, device: Optional[int] = None) -> "HashSet":
        h = C.c_void_p()
        check(lib.tbk_table_create_from_file(os.fsencode(path), default_device() if device is None else device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_keys(cls, keys, k: int, device: Optional[int] = None) -> "HashSet":
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        h = C.c_void_p()
        check(lib.tbk_table_create_from_keys(
            keys.ctypes.data, keys.size, k, default_device() if device is None else device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_device_keys(cls, d_keys: int, n: int, k: int, device: Optional[int] = None) -> "HashSet":
        h = C.c_void_p()
        check(lib.tbk_table_create_from_device_keys(
            C.c_void_p(d_keys), n, k, default_device() if device is None else device, C.byref(h)))
        return cls(h.value)

    # -- facts ----------------------------------------------------------------------------
    @property
    def num_kmers(self) -> int:
        """Number of list lines, duplicates included (reference: hash_set.num_kmers)."""
        return lib.tbk_table_num_kmers(self._h)

    @property
    def k(self) -> int:
        return lib.tbk_table_k(self._h)

    @property
    def device(self) -> int:
        return lib.tbk_table_device(self._h)

    @property
    def distinct(self) -> int:
        """Distinct keys in the list (hashes the list on its own on first use)."""
        n = C.c_uint64()
        check(lib.tbk_table_distinct(self._h, C.byref(n)))
        return n.value

    @property
    def nbytes(self) -> int:
        return lib.tbk_table_bytes(self._h)

    @property
    def device_keys(self) -> int:
        """Device pointer of the list's packed keys (num_kmers of them, read-only)."""
        return lib.tbk_table_device_keys(self._h)

    @property
    def contents(self) -> _Contents:
        return _Contents(self)

    @property
    def origin(self) -> str:
        """Where the keys came from: "keys" (the caller's, or the general host parser), "gpu-parser", "cache"."""
        return {0: "keys", 1: "gpu-parser", 2: "cache"}.get(lib.tbk_table_origin(self._h), "?")

    def keys(self) -> np.ndarray:
        """The packed keys, one per list line, copied to the host."""
        out = np.empty(self.num_kmers, dtype=np.uint64)
        check(lib.tbk_table_keys(self._h, out.ctypes.data, out.size))
        return out

    def contains(self, keys) -> np.ndarray:
        """Membership of raw packed keys (no canonicalisation)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(keys.size, dtype=np.uint8)
        check(lib.tbk_table_contains(self._h, keys.ctypes.data, keys.size, out.ctypes.data))
        return out.astype(bool)

    def close(self) -> None:
        if self._h is not None and self._h.value:
            lib.tbk_table_destroy(self._h)
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


# ---- the reference's five functions ------------------------------------------------------
def kmer_to_int(kmer: str) -> int:
    """Convert a kmer to integer format (reference kmers.py:80-82 -> c/kmers.c:50-72)."""
    raw = bytes(kmer, "utf-8")
    return lib.tbk_kmer_to_int(raw, C.c_ubyte(len(kmer)))


def reverse_complement(kmer: str) -> str:
    """Reverse complement a k-mer (reference kmers.py:89-93 -> c/kmers.c:74-93).

    As in the reference, the output starts as ``"x" * k`` and positions whose source
    base is not one of ACGT keep the ``x``.
    """
    k = len(kmer)
    out = C.create_string_buffer(b"x" * k, k + 1)
    lib.tbk_reverse_complement(bytes(kmer, "utf-8"), out, k)
    return out.raw[:k].decode("utf-8")


def create_kmer_hash_set(kmer_file_path: str) -> HashSet:
    """Read a list of k-mers (one per line) into a searchable set in HBM.

    Same contract as the reference (kmers.py:104-122): ``IOError`` when the path is not
    a file, a progress line on stderr, a handle for ``count_kmers_in_read``.
    """
    if not isfile(kmer_file_path):
        raise IOError(f"Specified file {kmer_file_path} does not exist or is not file.")
    print(f"Reading k-mers in {kmer_file_path}...", file=sys.stderr)
    hs = HashSet.from_file(kmer_file_path)
    print(f"Found {hs.num_kmers} {hs.k}-mers in {kmer_file_path} (in HBM on device {hs.device}).", file=sys.stderr)
    return hs


def parse_kmer_list(kmer_file_path: str) -> Tuple[np.ndarray, int]:
    """Host-only: (packed keys, k) of a k-mer list file, exactly what ``create_kmer_hash_set``
    places in HBM (the reference's getline rules, c/kmers.c:124-146,204-221)."""
    keys, n, k = _lib._u64p(), C.c_uint64(), C.c_int()
    check(lib.tbk_list_parse_file(os.fsencode(kmer_file_path), C.byref(keys), C.byref(n), C.byref(k)))
    try:
        out = np.ctypeslib.as_array(keys, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.uint64)
    finally:
        lib.tbk_list_free(keys)
    return out, k.value


def count_kmers_in_read(read: str, kmers_hap_a: HashSet, kmers_hap_b: HashSet) -> Tuple[int, int]:
    """Count the k-mers of one read found in each of two sets (reference kmers.py:125-154).

    Returns ``(count_a, count_b)``; a k-mer present in both sets counts for A only.  One
    kernel launch per call: use ``Classifier`` for throughput.
    """
    raw = read.encode("utf-8")
    ca, cb = C.c_int(), C.c_int()
    check(lib.tbk_count_kmers_in_read(raw, len(raw), kmers_hap_a._h, kmers_hap_b._h, C.byref(ca), C.byref(cb)))
    return ca.value, cb.value


def get_number_kmers_in_set(kmer_hash_set: HashSet) -> int:
    """Look up the number of k-mers in a hash set (reference kmers.py:157-159)."""
    return kmer_hash_set.contents.num_kmers


# ---- batch interface ---------------------------------------------------------------------
def pack_reads(seqs: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate read sequences into the batch layout of the C-ABI:
    ``bases`` (uint8, back to back) and ``offsets`` (uint64, n+1)."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    raw = "".join(seqs).encode("utf-8")
    if len(raw) != int(offsets[-1]):
        # non-ASCII text: lengths must be byte lengths
        enc = [s.encode("utf-8") for s in seqs]
        lens = np.fromiter((len(b) for b in enc), dtype=np.uint64, count=len(enc))
        np.cumsum(lens, out=offsets[1:])
        raw = b"".join(enc)
    bases = np.frombuffer(raw, dtype=np.uint8)
    return bases, offsets


class PackedBatch:
    """A read batch in the packed transfer format (include/tbk.h): 2-bit code words, the exceptions
    (chunks holding a byte outside ACGT or positions past the end) and the reads' offsets."""

    def __init__(self, codes, exc_chunk, exc_mask, offsets):
        self.codes, self.exc_chunk, self.exc_mask, self.offsets = codes, exc_chunk, exc_mask, offsets

    @property
    def n_reads(self) -> int:
        return self.offsets.size - 1

    @property
    def nbytes(self) -> int:
        """Bytes that cross PCIe for this batch."""
        return self.codes.nbytes + self.exc_chunk.nbytes + self.exc_mask.nbytes + self.offsets.nbytes


def pack_bases(bases: np.ndarray, offsets: np.ndarray, pinned: bool = True) -> PackedBatch:
    """Pack a batch's ASCII bases on the host (all host threads; AVX2 when the CPU has it)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    total = int(offsets[-1])
    n_chunks = int(lib.tbk_packed_chunks(total))
    alloc = pinned_empty if pinned else (lambda shape, dt: np.empty(shape, dtype=dt))
    codes = alloc((max(n_chunks, 1),), np.uint32)
    n_exc = C.c_uint64()
    cap = 1024
    while True:
        ec, em = np.empty(cap, dtype=np.uint32), np.empty(cap, dtype=np.uint16)
        rc = lib.tbk_pack_bases(bases.ctypes.data, total, codes.ctypes.data, ec.ctypes.data, em.ctypes.data, cap, C.byref(n_exc))
        if rc == _lib.TBK_ERR_NOMEM and n_exc.value > cap:
            cap = n_exc.value
            continue
        check(rc)
        break
    n = n_exc.value
    exc_chunk, exc_mask = alloc((max(n, 1),), np.uint32)[:n], alloc((max(n, 1),), np.uint16)[:n]
    exc_chunk[:] = ec[:n]
    exc_mask[:] = em[:n]
    if pinned:
        po = pinned_empty(offsets.shape, np.uint64)
        po[:] = offsets
        offsets = po
    return PackedBatch(codes[:n_chunks], exc_chunk, exc_mask, offsets)


class Options:
    """How a classifier is built (include/tbk.h: ``tbk_options``).  The reference configures itself through argparse alone
    (classify_by_kmers.py:14-54); this library's choices - which layout the paired table takes, the sampling rule, loads,
    the memory budget - are arguments too: ``Options(entries=1, entry_load=0.64)``, ``Options.layout("short_keys")``.
    No option changes a result.  ``Options.from_env()`` overlays the ``TBK_*`` variables of the process environment - the
    command-line tools' fallback, and what ``Classifier`` / ``MultiClassifier`` do when no options are given; the library's
    constructors themselves never read the environment, so two classifiers of one process can be built differently at the
    same time."""

    LAYOUTS = {
        "auto": {},
        "keys": {"short_keys": 0, "entries": 0, "full_keys": 0},
        "keys_front": {"short_keys": 0, "entries": 0, "full_keys": 0, "front": 1},
        "keys_whole_lines": {"short_keys": 0, "entries": 0, "full_keys": 0, "front": 0},
        "full_keys": {"short_keys": 0, "entries": 0, "full_keys": 1},
        "entries": {"entries": 1, "wide_entries": 0},
        "wide_entries": {"entries": 1, "wide_entries": 1},
        "short_keys": {"short_keys": 1},
    }

    def __init__(self, **fields):
        self.c = _lib.tbk_options()
        lib.tbk_options_init(C.byref(self.c))
        self.update(**fields)

    def update(self, **fields) -> "Options":
        names = {f[0] for f in _lib.tbk_options._fields_} - {"size"}
        for name, value in fields.items():
            if name not in names:
                raise TypeError(f"tbk_options has no field {name!r}")
            setattr(self.c, name, value)
        return self

    @classmethod
    def layout(cls, name: str, **fields) -> "Options":
        return cls(**dict(cls.LAYOUTS[name], **fields))

    @classmethod
    def from_env(cls, **fields) -> "Options":
        self = cls()
        check(lib.tbk_options_from_env(C.byref(self.c)))
        return self.update(**fields)

    def __repr__(self):
        dflt = _lib.tbk_options()
        lib.tbk_options_init(C.byref(dflt))
        diff = {f[0]: getattr(self.c, f[0]) for f in _lib.tbk_options._fields_ if getattr(self.c, f[0]) != getattr(dflt, f[0])}
        return f"Options({', '.join(f'{k}={v}' for k, v in diff.items())})"


def _options_ptr(options):
    """ctypes pointer for the *_opts calls: the given Options, or - none given - the TBK_* environment's (the fallback)"""
    if not _lib.HAS_OPTIONS:
        return None
    if options is None:
        options = Options.from_env()
    return C.byref(options.c)


class Classifier:
    """The batch hot path: per-read (hapA, hapB) k-mer hit counts for many reads at once.

    Replaces the per-read Python loop of the reference driver
    (classify_by_kmers.py:99-102).  ``submit``/``wait`` keep up to ``depth`` batches in
    flight so the host-to-device copy of the next batch overlaps the kernel of the
    current one.
    """

    def __init__(self, kmers_hap_a: HashSet, kmers_hap_b: HashSet, _handle: Optional[int] = None, options: Optional[Options] = None):
        if _handle is None:
            h = C.c_void_p()
            if _lib.HAS_OPTIONS:
                check(lib.tbk_classifier_create_opts(kmers_hap_a._h, kmers_hap_b._h, _options_ptr(options), C.byref(h)))
            else:
                check(lib.tbk_classifier_create(kmers_hap_a._h, kmers_hap_b._h, C.byref(h)))
        else:  # one of tbk_classifier_create_multi's classifiers
            h = C.c_void_p(_handle)
        self._h = h
        self._a, self._b = kmers_hap_a, kmers_hap_b  # keep the tables alive
        self._keep = {}
        self.device = lib.tbk_classifier_device(h)

    @property
    def depth(self) -> int:
        return lib.tbk_stream_depth(self._h)

    def stats(self) -> dict:
        """Distinct keys per list, bucket lines and bytes of the paired table in HBM."""
        da, db, nb, by = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.tbk_classifier_stats(self._h, C.byref(da), C.byref(db), C.byref(nb), C.byref(by)))
        w, m, o = C.c_int(), C.c_int(), C.c_int()
        check(lib.tbk_classifier_layout(self._h, C.byref(w), C.byref(m), C.byref(o)))
        sh = C.c_uint64()
        check(lib.tbk_classifier_shared_keys(self._h, C.byref(sh)))
        nb_, past = C.c_int(), C.c_uint64()
        check(lib.tbk_classifier_build_info(self._h, C.byref(nb_), C.byref(past)))
        front, behind = C.c_int(), C.c_uint64()
        if hasattr(lib, "tbk_classifier_front"):
            check(lib.tbk_classifier_front(self._h, C.byref(front), C.byref(behind)))
        ent, ea, eb = C.c_int(), C.c_uint64(), C.c_uint64()
        if hasattr(lib, "tbk_classifier_entries"):
            check(lib.tbk_classifier_entries(self._h, C.byref(ent), C.byref(ea), C.byref(eb)))
        return {"entry_layout": ent.value in (1, 2), "wide_entries": ent.value == 2, "short_keys": ent.value == 3, "full_keys": ent.value == 4, "entries_a": ea.value, "entries_b": eb.value, "shared_keys": sh.value, "layout_builds": nb_.value, "keys_past_half": past.value, "front_layout": bool(front.value), "keys_behind_front": behind.value, "distinct_a": da.value, "distinct_b": db.value, "n_buckets": nb.value, "table_bytes": by.value,
                "minimizer_w": w.value, "minimizer_m": m.value, "span_offset": o.value,
                "sampling_t": lib.tbk_classifier_sampling_t(self._h)}

    def verified(self) -> dict:
        """What the constructor's own check did (``tbk_options.verify_build``: by default every table built by inserts that
        merge keys - entries, wide entries - is asked for every line of both lists before it is handed out, on every device):
        {"lines": list lines looked up again (0: not checked), "seconds"}.  A table that answers a line wrongly is never
        returned - the constructor fails."""
        lines, sec = C.c_uint64(), C.c_double()
        if not hasattr(lib, "tbk_classifier_verified"):
            return {"lines": 0, "seconds": 0.0}
        check(lib.tbk_classifier_verified(self._h, C.byref(lines), C.byref(sec)))
        return {"lines": lines.value, "seconds": sec.value}

    def table_id(self) -> tuple:
        """(where the table lies - equal ids share one table -, how it was made: 0 built first or shared, 1 a copy of a finished
        table, 2 built again on this device from the lists in the first one's geometry)"""
        tid, rep = C.c_uint64(), C.c_int()
        check(lib.tbk_classifier_table_id(self._h, C.byref(tid), C.byref(rep)))
        return tid.value, rep.value

    def verify(self, kmers_hap_a: Optional[HashSet] = None, kmers_hap_b: Optional[HashSet] = None) -> dict:
        """Every line of both lists through the finished table, against the lists' standalone tables of verbatim keys
        (``tbk_classifier_verify``): {"lines", "count_a", "count_b", "bad_lines", "first_bad"}; raises nothing - the caller decides."""
        a, b = kmers_hap_a or self._a, kmers_hap_b or self._b
        out = (C.c_uint64 * 5)()
        check(lib.tbk_classifier_verify(self._h, a._h, b._h, out))
        return {"lines": out[0], "count_a": out[1], "count_b": out[2], "bad_lines": out[3], "first_bad": None if out[4] == 2 ** 64 - 1 else out[4]}

    def classify_batch(self, bases: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        counts = np.zeros((n, 2), dtype=np.int32)
        check(lib.tbk_classify_batch(self._h, bases.ctypes.data, offsets.ctypes.data, n, counts.ctypes.data))
        return counts

    def classify_reads(self, seqs: Sequence[str]) -> np.ndarray:
        return self.classify_batch(*pack_reads(seqs))

    def count_read(self, read: str) -> Tuple[int, int]:
        """One read, one kernel launch (``tbk_classifier_count_read``): what ``count_kmers_in_read`` - the reference's per-read
        call, kmers.py:125-154 - does on its cached classifier."""
        raw = read.encode("utf-8")
        ca, cb = C.c_int(), C.c_int()
        check(lib.tbk_classifier_count_read(self._h, raw, len(raw), C.byref(ca), C.byref(cb)))
        return ca.value, cb.value

    def submit(self, bases: np.ndarray, offsets: np.ndarray) -> int:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        counts = np.zeros((n, 2), dtype=np.int32)
        ticket = C.c_uint64()
        check(lib.tbk_stream_submit(self._h, bases.ctypes.data, offsets.ctypes.data, n, counts.ctypes.data, C.byref(ticket)))
        self._keep[ticket.value] = (bases, offsets, counts)
        return ticket.value

    def submit_packed(self, packed: PackedBatch) -> int:
        """Submit a batch packed ahead of time (``pack_bases``): 0.25 bytes per base cross PCIe."""
        counts = np.zeros((packed.n_reads, 2), dtype=np.int32)
        ticket = C.c_uint64()
        check(lib.tbk_stream_submit_packed(self._h, packed.codes.ctypes.data, packed.exc_chunk.ctypes.data, packed.exc_mask.ctypes.data,
                                           packed.exc_chunk.size, packed.offsets.ctypes.data, packed.n_reads, counts.ctypes.data, C.byref(ticket)))
        self._keep[ticket.value] = (packed, None, counts)
        return ticket.value

    @property
    def packed_transfer(self) -> bool:
        """Whether ``submit`` / ``classify_batch`` pack host batches before the copy (default; TBK_PACKED_H2D=0 turns it off)."""
        return bool(lib.tbk_classifier_transfer(self._h))

    @packed_transfer.setter
    def packed_transfer(self, on: bool) -> None:
        check(lib.tbk_classifier_set_transfer(self._h, int(bool(on))))

    def submit_batch(self, batch) -> int:
        """Submit a ``seq.Batch`` (its bases already lie in pinned memory: no staging copy)."""
        bases_ptr, off_ptr = batch.pointers()
        counts = np.zeros((batch.n_reads, 2), dtype=np.int32)
        ticket = C.c_uint64()
        check(lib.tbk_stream_submit(self._h, C.c_void_p(bases_ptr), C.c_void_p(off_ptr), batch.n_reads,
                                    counts.ctypes.data, C.byref(ticket)))
        self._keep[ticket.value] = (batch, None, counts)
        return ticket.value

    def wait(self, ticket: int) -> np.ndarray:
        check(lib.tbk_stream_wait(self._h, ticket))
        return self._keep.pop(ticket)[2]

    wait_ticket = wait

    # device-resident form (bench.py; inputs already in HBM)
    def classify_device(self, d_bases: int, d_offsets: int, n_reads: int, total_bases: int, d_counts: int) -> None:
        check(lib.tbk_classify_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, total_bases,
                                      C.c_void_p(d_counts)))

    def submit_device(self, d_bases: int, d_offsets: int, n_reads: int, total_bases: int, counts: np.ndarray) -> int:
        """Ticketed form of ``classify_device``: ``counts`` (int32 [n_reads, 2], ideally a view
        of pinned memory from ``pinned_empty``) is filled when ``wait_ticket`` returns."""
        ticket = C.c_uint64()
        check(lib.tbk_stream_submit_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, total_bases,
                                           counts.ctypes.data, C.byref(ticket)))
        self._keep[ticket.value] = (None, None, counts)
        return ticket.value

    def sync(self) -> None:
        check(lib.tbk_classifier_sync(self._h))

    def kernel_timing(self, on: bool) -> None:
        check(lib.tbk_kernel_timing_enable(self._h, int(on)))

    def kernel_timing_read(self) -> Tuple[int, float]:
        n, ms = C.c_uint64(), C.c_double()
        check(lib.tbk_kernel_timing_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_timing_read2(self) -> Tuple[int, float, float]:
        """Probes since enable, their summed milliseconds (pass index + both probe kernels) and those of
        the single-read kernel alone - on long reads the dominant kernel."""
        n, ms, single = C.c_uint64(), C.c_double(), C.c_double()
        check(lib.tbk_kernel_timing_read2(self._h, C.byref(n), C.byref(ms), C.byref(single)))
        return n.value, ms.value, single.value

    def calibrate(self) -> float:
        """Random 64-byte line reads per second over this table where it lies in HBM (a diagnostic)."""
        v = C.c_double()
        check(lib.tbk_classifier_calibrate(self._h, C.byref(v)))
        return v.value

    def calibrate_pairs(self, inflight: int = 4, waves_per_simd: int = 8, n_lines: int = 1 << 28) -> float:
        """The same in the entry kernels' own request shape (two lanes x 16 bytes of a line, one-wave blocks): random lines
        per second over this table - the ceiling of the window loop's line rate on this table, on this box, in this process."""
        v = C.c_double()
        check(lib.tbk_classifier_calibrate_pairs(self._h, inflight, waves_per_simd, n_lines, C.byref(v)))
        return v.value

    def last_passes(self) -> Tuple[int, int]:
        """(passes, passes that touch more than one read) of the most recent probe: 2048 window starts each; the
        latter went through the two-read or the multi-read kernel, the rest through the single-read kernel."""
        n, m = C.c_uint64(), C.c_uint64()
        check(lib.tbk_classifier_last_passes(self._h, C.byref(n), C.byref(m)))
        return n.value, m.value

    def close(self) -> None:
        if self._h is not None and self._h.value:
            lib.tbk_classifier_destroy(self._h)
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


def visible_devices() -> list:
    """Device indices a multi-device run uses: TBK_DEVICES ("0,1,2", a device may repeat), else
    every visible device."""
    spec = os.environ.get("TBK_DEVICES", "").strip()
    if spec:
        return [int(x) for x in spec.split(",") if x.strip() != ""]
    return list(range(_lib.device_count()))


class MultiClassifier:
    """Reads dealt over several devices behind ``Classifier``'s interface (tables replicated,
    no collective; the reference's loop, classify_by_kmers.py:99-102, has no cross-read state).

    A thin wrapper over the library's ``tbk_pipeline``: one feeder thread and one stream ring per entry of
    ``devices`` (a device may repeat), one ordered queue.  ``submit`` only queues a batch and returns a
    ticket; the next feeder whose ring has room takes it.  ``wait(ticket)`` returns that batch's counts
    whichever device computed them, so a caller that waits in submission order gets its results in input
    order.  ``depth`` batches may be in flight on the devices; ``depth + len(devices)`` may be submitted
    and not yet waited for.  Nothing per batch happens in Python.
    """

    def __init__(self, kmers_hap_a: HashSet, kmers_hap_b: HashSet, devices: Optional[Sequence[int]] = None, options: Optional[Options] = None):
        devices = list(visible_devices() if devices is None else devices)
        if not devices:
            raise _lib.TbkError(_lib.TBK_ERR_NO_DEVICE, "no HIP device visible; there is no CPU fallback")
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        if _lib.HAS_OPTIONS:
            check(lib.tbk_pipeline_create_opts(kmers_hap_a._h, kmers_hap_b._h, arr, len(devices), _options_ptr(options), C.byref(h)))
        else:
            check(lib.tbk_pipeline_create(kmers_hap_a._h, kmers_hap_b._h, arr, len(devices), C.byref(h)))
        self._h = h
        self._a, self._b = kmers_hap_a, kmers_hap_b  # keep the tables alive
        self._keep = {}
        self._stubs = None
        self.devices = devices
        self.device = devices[0]

    @classmethod
    def from_classifiers(cls, classifiers, ring_depth: Optional[int] = None) -> "MultiClassifier":
        """The same queue and feeder threads over classifier-like Python objects (``submit(bases, offsets)
        -> ticket``, ``wait(ticket) -> counts``): the library's testing hook, no GPU needed."""
        self = cls.__new__(cls)
        parts = list(classifiers)
        self._stubs = parts
        self._keep = {}
        self._a = self._b = None
        self.devices = [getattr(p, "device", None) for p in parts]
        self.device = self.devices[0]
        pending = {}  # (slot, stub ticket) -> (counts pointer, n_reads)

        def on_submit(user, slot, bases, offsets, n_reads, counts, ticket):
            try:
                off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), (n_reads + 1,))
                total = int(off[n_reads])
                b = np.ctypeslib.as_array(C.cast(bases, C.POINTER(C.c_uint8)), (total,)) if total else np.zeros(0, dtype=np.uint8)
                t = parts[slot].submit(b, off)
                pending[(slot, t)] = (counts, n_reads)
                ticket[0] = t
                return 0
            except Exception as exc:  # reported through the job's status
                self._stub_error = exc
                return _lib.TBK_ERR_STATE

        def on_wait(user, slot, ticket):
            try:
                got = np.ascontiguousarray(parts[slot].wait(ticket), dtype=np.int32)
                ptr, n = pending.pop((slot, ticket))
                if n:
                    C.memmove(ptr, got.ctypes.data, n * 8)
                return 0
            except Exception as exc:
                self._stub_error = exc
                return _lib.TBK_ERR_STATE

        sub_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64))
        wait_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64)
        self._callbacks = (sub_t(on_submit), wait_t(on_wait))  # kept alive with the pipeline
        lib.tbk_pipeline_create_test_.restype = C.c_int
        lib.tbk_pipeline_create_test_.argtypes = [C.c_int, C.c_int, sub_t, wait_t, C.c_void_p, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        depth = ring_depth if ring_depth is not None else min(getattr(p, "depth", 3) for p in parts)
        check(lib.tbk_pipeline_create_test_(len(parts), depth, self._callbacks[0], self._callbacks[1], None, C.byref(h)))
        self._h = h
        return self

    @property
    def depth(self) -> int:
        return lib.tbk_pipeline_depth(self._h)

    @property
    def dealt(self) -> list:
        """Batches each ring has taken so far."""
        n = lib.tbk_pipeline_devices(self._h)
        out = (C.c_uint64 * n)()
        check(lib.tbk_pipeline_batches(self._h, out, n))
        return [int(x) for x in out]

    def _part(self, slot: int) -> "Classifier":
        c = Classifier.__new__(Classifier)
        c._h = C.c_void_p(lib.tbk_pipeline_classifier(self._h, slot))
        c._a, c._b, c._keep = self._a, self._b, {}
        c.device = lib.tbk_classifier_device(c._h)
        c.close = lambda: None  # the pipeline owns it
        return c

    def stats(self) -> dict:
        st = dict(self._stubs[0].stats() if self._stubs is not None else self._part(0).stats())
        st["devices"] = list(self.devices)
        st["table_bytes_total"] = st.get("table_bytes", 0) * len(set(self.devices))  # the rings of one device share its table
        return st

    def kernel_timing(self, on: bool) -> None:
        for i in range(len(self.devices)):
            self._part(i).kernel_timing(on)

    def kernel_timing_read(self):
        """Summed over the rings: probes, their milliseconds, the single-read kernel's milliseconds."""
        tot = [0, 0.0, 0.0]
        for i in range(len(self.devices)):
            for j, v in enumerate(self._part(i).kernel_timing_read2()):
                tot[j] += v
        return tuple(tot)

    def submit(self, bases: np.ndarray, offsets: np.ndarray) -> int:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        counts = np.zeros((n, 2), dtype=np.int32)
        ticket = C.c_uint64()
        check(lib.tbk_pipeline_submit(self._h, bases.ctypes.data, offsets.ctypes.data, n, counts.ctypes.data, C.byref(ticket)))
        self._keep[ticket.value] = (bases, offsets, counts)
        return ticket.value

    def submit_packed(self, packed: PackedBatch, counts: Optional[np.ndarray] = None) -> int:
        if counts is None:
            counts = np.zeros((packed.n_reads, 2), dtype=np.int32)
        ticket = C.c_uint64()
        check(lib.tbk_pipeline_submit_packed(self._h, packed.codes.ctypes.data, packed.exc_chunk.ctypes.data, packed.exc_mask.ctypes.data,
                                             packed.exc_chunk.size, packed.offsets.ctypes.data, packed.n_reads, counts.ctypes.data, C.byref(ticket)))
        self._keep[ticket.value] = (packed, None, counts)
        return ticket.value

    def submit_batch(self, batch) -> int:
        """Submit a ``seq.Batch``: in the packed transfer format when the reader made it, else its ASCII bases."""
        counts = np.zeros((batch.n_reads, 2), dtype=np.int32)
        ticket = C.c_uint64()
        bases_ptr, off_ptr = batch.pointers()
        packed = batch.packed_pointers() if self._stubs is None else None
        if packed is not None:
            codes, exc_chunk, exc_mask, n_exc = packed
            check(lib.tbk_pipeline_submit_packed(self._h, C.c_void_p(codes), C.c_void_p(exc_chunk), C.c_void_p(exc_mask), n_exc,
                                                 C.c_void_p(off_ptr), batch.n_reads, counts.ctypes.data, C.byref(ticket)))
        else:
            check(lib.tbk_pipeline_submit(self._h, C.c_void_p(bases_ptr), C.c_void_p(off_ptr), batch.n_reads, counts.ctypes.data, C.byref(ticket)))
        self._keep[ticket.value] = (batch, None, counts)
        return ticket.value

    def wait(self, ticket: int) -> np.ndarray:
        try:
            check(lib.tbk_pipeline_wait(self._h, ticket, None))
        except _lib.TbkError:
            exc, self._stub_error = getattr(self, "_stub_error", None), None
            if exc is not None:
                raise exc
            raise
        return self._keep.pop(ticket)[2]

    wait_ticket = wait

    def classify_batch(self, bases: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        return self.wait(self.submit(bases, offsets))

    def classify_reads(self, seqs: Sequence[str]) -> np.ndarray:
        return self.classify_batch(*pack_reads(seqs))

    def classify_file(self, reads_path: str, num_kmers_a: int, num_kmers_b: int, out_names: Sequence[str], gzip_output: bool,
                      gzip_level: int = -1, tsv_fd: int = 1, batch_bases: int = 0, batch_reads: int = 0) -> dict:
        """The whole read / classify / write loop of classify-by-kmers on native threads (``tbk_classify_file``);
        the TSV goes to ``tsv_fd``.  Returns the run's statistics."""

        class Stats(C.Structure):
            _fields_ = [("reads", C.c_uint64), ("bases", C.c_uint64), ("batches", C.c_uint64), ("read_s", C.c_double),
                        ("gpu_wait_s", C.c_double), ("write_s", C.c_double), ("total_s", C.c_double), ("gzip_encoder", C.c_int32), ("_pad", C.c_int32)]

        st = Stats()
        a, b, u = [os.fsencode(n) for n in out_names]
        check(lib.tbk_classify_file(self._h, os.fsencode(reads_path), num_kmers_a, num_kmers_b, a, b, u, int(bool(gzip_output)), gzip_level,
                                    tsv_fd, batch_bases, batch_reads, C.byref(st)))
        out = {name: getattr(st, name) for name, _ in Stats._fields_ if name != "_pad"}
        out["gzip_encoder"] = {0: None, 1: "host", 2: "device"}.get(st.gzip_encoder)   # who coded the bins' gzip members
        return out

    def sync(self) -> None:
        pass  # every ticket that was waited for is complete; there is nothing else in flight to wait for

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            lib.tbk_pipeline_destroy(h)
        if self._stubs is not None:
            for p in self._stubs:
                p.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


Pipeline = MultiClassifier


class _PinnedOwner:
    """Frees a tbk_host_alloc block when the array that views it is collected."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            lib.tbk_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def host_threads() -> int:
    """Host threads worth starting (hardware threads within the affinity mask and the cgroup quota)."""
    return int(lib.tbk_host_threads())


def device_mem_info(device: Optional[int] = None) -> Tuple[int, int]:
    """(free, total) bytes of HBM on the device."""
    free, total = C.c_uint64(), C.c_uint64()
    check(lib.tbk_device_mem_info(default_device() if device is None else device, C.byref(free), C.byref(total)))
    return free.value, total.value


class KmerCounter:
    """Counting table of canonical k-mers in HBM: the database `kmc` builds for the reference's
    find-unique-kmers step (find_unique_kmers.py:62-103), with the histogram, subtraction and dump
    `kmc_tools` / `kmc_dump` provide (find_unique_kmers.py:123-129,186-194,218-225)."""

    def __init__(self, k: int, capacity: int, device: Optional[int] = None):
        self._h = C.c_void_p()
        self.k = k
        self.device = default_device() if device is None else device
        check(lib.tbk_counter_create(k, capacity, self.device, C.byref(self._h)))

    def add_reads(self, reads: Sequence[str]) -> None:
        bases, offsets = pack_reads(reads)
        self.add(bases, offsets)

    def add(self, bases: np.ndarray, offsets: np.ndarray) -> None:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        check(lib.tbk_counter_add_batch(self._h, bases.ctypes.data, offsets.ctypes.data, offsets.size - 1))

    def add_batch(self, batch) -> None:
        """Count a ``seq.Batch`` straight from its (pinned) buffers."""
        bases_ptr, off_ptr = batch.pointers()
        check(lib.tbk_counter_add_batch(self._h, C.c_void_p(bases_ptr), C.c_void_p(off_ptr), batch.n_reads))

    def add_device(self, d_bases: int, d_offsets: int, n_reads: int, total_bases: int) -> None:
        check(lib.tbk_counter_add_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, total_bases))

    def kernel_timing(self, reset: bool = True) -> Tuple[int, int, float]:
        """(launches, window starts, total ms) of the counting kernel since the last reset."""
        n, w, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        check(lib.tbk_counter_kernel_timing(self._h, C.byref(n), C.byref(w), C.byref(ms), int(reset)))
        return n.value, w.value, ms.value

    def histogram(self) -> np.ndarray:
        """hist[c], c = 1..255: distinct k-mers whose counter (capped at 255) is c; hist[0]: all."""
        hist = np.zeros(256, dtype=np.uint64)
        check(lib.tbk_counter_histogram(self._h, hist.ctypes.data_as(C.POINTER(C.c_uint64))))
        return hist

    def stats(self) -> dict:
        v = [C.c_uint64() for _ in range(4)]
        check(lib.tbk_counter_stats(self._h, *[C.byref(x) for x in v]))
        d = C.c_uint64()
        check(lib.tbk_counter_distinct(self._h, C.byref(d)))
        return dict(zip(("n_slots", "table_bytes", "bases_added", "reads_added", "distinct"), [x.value for x in v] + [d.value]))

    def unique(self, other: "KmerCounter", min_count: int, max_count: int, out_path: str) -> int:
        """Write the k-mers this library saw at least twice, with a counter in [min_count,
        max_count], that `other` saw at most once; one per line, sorted.  Returns how many."""
        n = C.c_uint64()
        check(lib.tbk_counter_unique(self._h, other._h, min_count, max_count, os.fsencode(out_path), C.byref(n)))
        return n.value

    def close(self) -> None:
        if self._h is not None and self._h.value:
            h, self._h = self._h, None
            lib.tbk_counter_destroy(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """A numpy array backed by pinned host memory (hipHostMalloc through the C-ABI), so that
    async copies to/from the GPU need no staging copy.  Freed with the array."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dtype.itemsize, 1)
    ptr = lib.tbk_host_alloc(nbytes)
    if not ptr:
        raise MemoryError(_lib.last_error())
    buf = (C.c_char * nbytes).from_address(ptr)
    buf._tbk_owner = _PinnedOwner(ptr)  # the array keeps `buf` alive through .base
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def score_and_bin(counts: np.ndarray, num_kmers_a: int, num_kmers_b: int):
    """Scores and bins for a batch (classify_by_kmers.py:57-77,104-115), float64, in the
    reference's operation order.  Returns (score_a, score_b, bins) with bins a bytes object
    over b"ABU"."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n = counts.shape[0]
    sa = np.zeros(n, dtype=np.float64)
    sb = np.zeros(n, dtype=np.float64)
    bins = C.create_string_buffer(n + 1)
    check(lib.tbk_score_and_bin(counts.ctypes.data, n, num_kmers_a, num_kmers_b, sa.ctypes.data, sb.ctypes.data, bins))
    return sa, sb, bins.raw[:n]
