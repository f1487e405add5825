"""ctypes bindings for the oracle restatement and (optionally) the real reference build.

Test infrastructure only — see ``oracle/__init__.py``.
"""
import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkmers_oracle.so")
_REF = os.path.join(_HERE, "_ref", "kmers_ref.so")


def build(quiet: bool = True) -> None:
    """Compile the restatement (and oracle/_ref when the reference mount exists)."""
    subprocess.run(
        ["make", "-C", _HERE, "all"],
        check=True,
        stdout=subprocess.DEVNULL if quiet else None,
    )


def have_ref() -> bool:
    return os.path.isfile(_REF)


class _Table:
    """Owning handle for an ``orc_table*``."""

    def __init__(self, lib, ptr):
        if not ptr:
            raise ValueError("oracle: table could not be built")
        self._lib, self.ptr = lib, ptr

    @property
    def num_kmers(self) -> int:
        return self._lib.orc_table_num_kmers(self.ptr)

    @property
    def hash_size(self) -> int:
        return self._lib.orc_table_hash_size(self.ptr)

    @property
    def k(self) -> int:
        return self._lib.orc_table_k(self.ptr)

    def keys(self) -> np.ndarray:
        """The stored keys (one per list line, duplicates included), sorted."""
        out = np.empty(max(1, self.hash_size), dtype=np.uint64)
        self._lib.orc_table_keys.restype = C.c_uint64
        self._lib.orc_table_keys.argtypes = [C.c_void_p, C.c_void_p]
        n = self._lib.orc_table_keys(self.ptr, out.ctypes.data)
        return np.sort(out[:n])

    def __del__(self):
        try:
            self._lib.orc_table_free(self.ptr)
        except Exception:
            pass


class Oracle:
    def __init__(self, path: str = _LIB):
        if not os.path.isfile(path):
            build()
        lib = C.CDLL(path)
        vp, u64p, i32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
        lib.orc_kmer_to_int.argtypes = [C.c_char_p, C.c_int]
        lib.orc_kmer_to_int.restype = C.c_uint64
        lib.orc_reverse_complement.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        lib.orc_hash.argtypes = [C.c_uint64]
        lib.orc_hash.restype = C.c_uint32
        lib.orc_table_from_file.argtypes = [C.c_char_p]
        lib.orc_table_from_file.restype = vp
        lib.orc_table_from_keys.argtypes = [u64p, C.c_uint64, C.c_int]
        lib.orc_table_from_keys.restype = vp
        lib.orc_table_from_keys_mt.argtypes = [u64p, C.c_uint64, C.c_int, C.c_int]
        lib.orc_table_from_keys_mt.restype = vp
        lib.orc_table_free.argtypes = [vp]
        for f in (lib.orc_table_num_kmers, lib.orc_table_hash_size):
            f.argtypes, f.restype = [vp], C.c_uint64
        lib.orc_table_k.argtypes, lib.orc_table_k.restype = [vp], C.c_int
        lib.orc_count_kmers_in_read.argtypes = [
            C.c_char_p, C.c_int64, vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.orc_count_batch.argtypes = [
            C.c_void_p, u64p, C.c_uint64, vp, vp, C.c_int, C.c_int, i32p]
        lib.orc_count_batch_fast.argtypes = [C.c_void_p, u64p, C.c_uint64, vp, vp, C.c_int, i32p]
        lib.orc_score_and_bin.argtypes = [
            i32p, C.c_uint64, C.c_uint64, C.c_uint64,
            C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p]
        lib.orc_kmer_histogram.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int, C.c_uint64, u64p]
        self.lib = lib

    def kmer_histogram(self, bases: np.ndarray, offsets: np.ndarray, k: int, expected_distinct: int) -> np.ndarray:
        """Canonical k-mer count histogram of a batch (find-unique-kmers step), single thread."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        slots = 1 << max(10, int(2 * expected_distinct - 1).bit_length())
        hist = np.zeros(256, dtype=np.uint64)
        rc = self.lib.orc_kmer_histogram(bases.ctypes.data, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), offsets.size - 1, k,
                                         slots, hist.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc:
            raise MemoryError("oracle k-mer table too small")
        return hist

    # -- unit-level API ---------------------------------------------------------------
    def kmer_to_int(self, kmer: str) -> int:
        b = kmer.encode()
        return self.lib.orc_kmer_to_int(b, len(b))

    def reverse_complement(self, kmer: str, fill: str = "x") -> str:
        b = kmer.encode()
        out = C.create_string_buffer((fill * len(b)).encode(), len(b) + 1)
        self.lib.orc_reverse_complement(b, out, len(b))
        return out.raw[: len(b)].decode()

    def hash(self, x: int) -> int:
        return self.lib.orc_hash(x)

    # -- tables -----------------------------------------------------------------------
    def table_from_file(self, path: str) -> _Table:
        return _Table(self.lib, self.lib.orc_table_from_file(path.encode()))

    def table_from_keys(self, keys: np.ndarray, k: int, threads: int = 1) -> _Table:
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        p = keys.ctypes.data_as(C.POINTER(C.c_uint64))
        if threads > 1:
            ptr = self.lib.orc_table_from_keys_mt(p, keys.size, k, threads)
        else:
            ptr = self.lib.orc_table_from_keys(p, keys.size, k)
        return _Table(self.lib, ptr)

    # -- hot path ---------------------------------------------------------------------
    def count_kmers_in_read(self, read, a: _Table, b: _Table, strict: bool = True) -> Tuple[int, int]:
        raw = read.encode() if isinstance(read, str) else bytes(read)
        ca, cb = C.c_int(), C.c_int()
        self.lib.orc_count_kmers_in_read(raw, len(raw), a.ptr, b.ptr, int(strict), C.byref(ca), C.byref(cb))
        return ca.value, cb.value

    def count_batch(self, bases: np.ndarray, offsets: np.ndarray, a: _Table, b: _Table,
                    strict: bool = True, threads: int = 1) -> np.ndarray:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        counts = np.zeros((n, 2), dtype=np.int32)
        self.lib.orc_count_batch(
            bases.ctypes.data, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, a.ptr, b.ptr,
            int(strict), threads, counts.ctypes.data_as(C.POINTER(C.c_int32)))
        return counts

    def count_batch_fast(self, bases: np.ndarray, offsets: np.ndarray, a: _Table, b: _Table, threads: int = 1) -> np.ndarray:
        """Fairness datum, not the reference's algorithm: rolling k-mers + threads."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        counts = np.zeros((n, 2), dtype=np.int32)
        self.lib.orc_count_batch_fast(bases.ctypes.data, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, a.ptr, b.ptr,
                                      threads, counts.ctypes.data_as(C.POINTER(C.c_int32)))
        return counts

    def score_and_bin(self, counts: np.ndarray, num_a: int, num_b: int):
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        n = counts.shape[0]
        sa, sb = np.zeros(n), np.zeros(n)
        bins = C.create_string_buffer(n + 1)
        self.lib.orc_score_and_bin(
            counts.ctypes.data_as(C.POINTER(C.c_int32)), n, num_a, num_b,
            sa.ctypes.data_as(C.POINTER(C.c_double)), sb.ctypes.data_as(C.POINTER(C.c_double)), bins)
        return sa, sb, bins.raw[:n].decode()


class _RefHashSet(C.Structure):
    # layout of the reference's struct (c/kmers.c:12-38), needed to read num_kmers back
    _fields_ = [("kmers", C.POINTER(C.c_uint64)), ("full", C.POINTER(C.c_ubyte)),
                ("hash_size", C.c_int), ("k", C.c_ubyte), ("num_kmers", C.c_int)]


class RefLib:
    """The real reference (oracle/_ref/kmers_ref.so), bound directly."""

    def __init__(self, path: str = _REF):
        lib = C.CDLL(path)
        hp = C.POINTER(_RefHashSet)
        lib.create_kmer_hash_set.argtypes, lib.create_kmer_hash_set.restype = [C.c_char_p], hp
        lib.count_kmers_in_read.argtypes = [C.c_char_p, hp, hp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.kmer_to_int.argtypes, lib.kmer_to_int.restype = [C.c_char_p, C.c_ubyte], C.c_uint64
        lib.reverse_complement.argtypes = [C.c_char_p, C.c_char_p, C.c_ubyte]
        lib.hash_function.argtypes, lib.hash_function.restype = [C.c_uint64], C.c_uint
        self.lib = lib

    def create_kmer_hash_set(self, path: str):
        return self.lib.create_kmer_hash_set(path.encode())

    @staticmethod
    def keys(hs) -> np.ndarray:
        """The keys a reference hash_set stores (its kmers[] where full[] is set), sorted."""
        c = hs.contents
        full = np.ctypeslib.as_array(c.full, shape=(c.hash_size,)).astype(bool)
        return np.sort(np.ctypeslib.as_array(c.kmers, shape=(c.hash_size,))[full].copy())

    def count_kmers_in_read(self, read: str, a, b) -> Tuple[int, int]:
        ca, cb = C.c_int(), C.c_int()
        self.lib.count_kmers_in_read(read.encode(), a, b, C.byref(ca), C.byref(cb))
        return ca.value, cb.value

    def kmer_to_int(self, kmer: str) -> int:
        return self.lib.kmer_to_int(kmer.encode(), len(kmer))

    def reverse_complement(self, kmer: str) -> str:
        out = C.create_string_buffer(b"x" * len(kmer), len(kmer) + 1)
        self.lib.reverse_complement(kmer.encode(), out, len(kmer))
        return out.raw[: len(kmer)].decode()

    def hash_function(self, x: int) -> int:
        return self.lib.hash_function(x)


_oracle: Optional[Oracle] = None


def load() -> Oracle:
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def load_ref() -> Optional[RefLib]:
    return RefLib() if have_ref() else None
