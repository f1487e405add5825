"""The full-membership sweep of a built paired table (tests/test_gpu_scale.py, ``bench.py --sweep``).

The reference stores every list line (``add_to_hash``, c/kmers.c:112-122) and finds every stored canonical key, and
nothing else (``kmer_in_hash_set``, c/kmers.c:245-268).  The tables of this library hold compressed and merged forms of
the keys - short keys, entries, wide entries, slot-pair orders - so at BASELINE's table sizes that claim is checked key
by key, through the ordinary probe path (``tbk_classifier_sweep_keys``, include/tbk.h):

  members       every key of hapA's list, laid out as a read of k bases (even index as it stands, odd index
                reverse-complemented), must count (1, 0); every key of hapB's list (0, 1) - or (1, 0) where hapA's list
                holds it too (hapA is asked first, c/kmers.c:291-294);
  long reads    the same keys, 512 to a read with an N between neighbours: the single-read and two-read kernels;
  non-members   as many keys that are in neither list must count (0, 0): BASELINE's uniform lists are a bijection of
                the index (tbk_synth_key), so indices [2N, 3N) are non-members by construction; for other lists random
                k-mers, with what they must count taken from the lists' standalone tables (verbatim 64-bit keys);
  near misses   every list key with ONE base substituted (canonicalised): shares m-mer, position and most flank bits
                with a member - what a compressed slot could confuse it with.  Expectation from the standalone tables.

Everything stays on the device; a leg returns sums and the number (and first index) of reads that differ.
"""
import ctypes as C
import time

from ._lib import check, lib

LONG_READ_KEYS = 512


def _sweep(cls, d_keys, n, k, per_read, expect=0, d_expect=None, chunk=0):
    out = (C.c_uint64 * 4)()
    check(lib.tbk_classifier_sweep_keys(cls._h, C.c_void_p(d_keys), n, k, per_read, expect, C.c_void_p(d_expect) if d_expect else None, chunk, out))
    return {"sum_a": out[0], "sum_b": out[1], "bad_reads": out[2], "first_bad": None if out[3] == 2 ** 64 - 1 else out[3]}


def full_membership_sweep(cls, list_a, list_b, d_keys_a, d_keys_b, n_a, n_b, k, device=0, uniform_seed=None, chunk=1 << 26,
                          legs=("members", "long_reads", "non_members", "near_misses")):
    """cls: the Classifier over (list_a, list_b) - HashSets whose keys also lie at d_keys_a / d_keys_b (device pointers,
    n_a / n_b canonical keys).  uniform_seed: the lists are keys [0, n_a) and [n_a, n_a + n_b) of tbk_synth_key(seed):
    indices beyond are non-members by construction.  Returns {"ok": bool, "legs": [...]}; a leg is ok when no read
    differs from what it must count and the sums are what the lists' sizes say."""
    st = cls.stats()
    shared = st["shared_keys"]
    records = []

    def dalloc(nbytes):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(device, max(16, nbytes), C.byref(p)))
        return p.value

    def leg(name, n, fn, want_a, want_b):
        t0 = time.time()
        r = fn()
        r.update(leg=name, keys=n, seconds=round(time.time() - t0, 2))
        r["ok"] = r["bad_reads"] == 0 and (want_a is None or r["sum_a"] == want_a) and (want_b is None or r["sum_b"] == want_b)
        r["want"] = [want_a, want_b]
        records.append(r)

    d_tmp, d_exp = dalloc(chunk * 8), dalloc(chunk)

    def chunked(src, n, make_keys, const_expect=None):
        """sweep keys produced chunk by chunk by make_keys(first, cn) -> device pointer, one read of k bases per key, against
        const_expect or the standalone tables' word"""
        tot = {"sum_a": 0, "sum_b": 0, "bad_reads": 0, "first_bad": None}
        for first in range(0, n, chunk):
            cn = min(chunk, n - first)
            d_k = make_keys(first, cn)
            if const_expect is None:
                check(lib.tbk_sweep_expectation_device(list_a._h, list_b._h, C.c_void_p(d_k), cn, C.c_void_p(d_exp)))
                r = _sweep(cls, d_k, cn, k, 1, 0, d_exp)
            else:
                r = _sweep(cls, d_k, cn, k, 1, const_expect)
            tot["sum_a"] += r["sum_a"]; tot["sum_b"] += r["sum_b"]; tot["bad_reads"] += r["bad_reads"]
            if tot["first_bad"] is None and r["first_bad"] is not None:
                tot["first_bad"] = first + r["first_bad"]
        return tot

    try:
        if "members" in legs:
            leg("members_hapA_k_base_reads", n_a, lambda: _sweep(cls, d_keys_a, n_a, k, 1, 1), n_a, 0)
            if shared == 0:
                leg("members_hapB_k_base_reads", n_b, lambda: _sweep(cls, d_keys_b, n_b, k, 1, 2), 0, n_b)
            else:  # hapB keys that hapA's list holds count for hapA: which ones, the standalone table of hapA's list says
                leg("members_hapB_k_base_reads", n_b, lambda: chunked(d_keys_b, n_b, lambda f, cn: d_keys_b + 8 * f), shared, n_b - shared)
        if "long_reads" in legs:
            leg("members_hapA_long_reads", n_a, lambda: _sweep(cls, d_keys_a, n_a, k, LONG_READ_KEYS, 1), n_a, 0)
            if shared == 0:
                leg("members_hapB_long_reads", n_b, lambda: _sweep(cls, d_keys_b, n_b, k, LONG_READ_KEYS, 2), 0, n_b)
            else:
                # hapB keys that hapA's list holds count for hapA: every long read must count exactly what ITS 512 keys say (per-key
                # expectations from the lists' standalone tables, summed per read on the device) - no allowance by count
                def b_long():
                    tot = {"sum_a": 0, "sum_b": 0, "bad_reads": 0, "first_bad": None}
                    step = max(LONG_READ_KEYS, chunk // LONG_READ_KEYS * LONG_READ_KEYS)   # whole reads per chunk
                    for first in range(0, n_b, step):
                        cn = min(step, n_b - first)
                        check(lib.tbk_sweep_expectation_device(list_a._h, list_b._h, C.c_void_p(d_keys_b + 8 * first), cn, C.c_void_p(d_exp)))
                        r = _sweep(cls, d_keys_b + 8 * first, cn, k, LONG_READ_KEYS, 0, d_exp)
                        tot["sum_a"] += r["sum_a"]; tot["sum_b"] += r["sum_b"]; tot["bad_reads"] += r["bad_reads"]
                        if tot["first_bad"] is None and r["first_bad"] is not None:
                            tot["first_bad"] = first // LONG_READ_KEYS + r["first_bad"]
                    tot["note"] = f"{shared} hapB keys are hapA's too: every read is checked against what its own {LONG_READ_KEYS} keys say"
                    return tot
                leg("members_hapB_long_reads", n_b, b_long, shared, n_b - shared)
        if "non_members" in legs:
            n_non = max(n_a, n_b)
            if uniform_seed is not None:
                def synth(first, cn):
                    check(lib.tbk_synth_keys_device(device, uniform_seed, n_a + n_b + first, cn, k, C.c_void_p(d_tmp)))
                    return d_tmp
                leg("non_members_next_indices_of_the_bijection", n_non, lambda: chunked(None, n_non, synth, const_expect=0), 0, 0)
            else:
                def synth(first, cn):
                    check(lib.tbk_synth_keys_device(device, 0x0DDBA11, first, cn, k, C.c_void_p(d_tmp)))
                    return d_tmp
                leg("non_members_random_kmers", n_non, lambda: chunked(None, n_non, synth), None, None)
        if "near_misses" in legs:
            for name, d_src, n in (("near_misses_of_hapA_keys", d_keys_a, n_a), ("near_misses_of_hapB_keys", d_keys_b, n_b)):
                def mutate(first, cn, d_src=d_src):
                    check(lib.tbk_synth_mutate_keys_device(device, C.c_void_p(d_src + 8 * first), first, cn, k, 0x5EED0005, C.c_void_p(d_tmp)))
                    return d_tmp
                leg(name, n, lambda: chunked(None, n, mutate), None, None)
    finally:
        check(lib.tbk_device_free(device, C.c_void_p(d_tmp)))
        check(lib.tbk_device_free(device, C.c_void_p(d_exp)))
    return {"ok": all(r["ok"] for r in records), "shared_keys": shared, "legs": records}
