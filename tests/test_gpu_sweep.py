"""The full-membership sweep (trio_binning_amd/sweep.py; include/tbk.h "the full-membership sweep") at small sizes, in
every layout of the paired table: every list key must answer for its list, hapB keys that hapA holds for hapA
(c/kmers.c:291-294), non-members and near misses must count what the standalone tables say (c/kmers.c:245-268) - and the
sweep itself must notice a table that answers wrongly (the controls at the end).  tests/test_gpu_scale.py runs the same
sweep at BASELINE's table sizes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAYOUTS = {  # (name of Options.layout, further tbk_options fields)
    "key_front": ("keys_front", {}),
    "key_whole_lines": ("keys_whole_lines", {}),
    "entries": ("entries", {}),
    "entries_crowded": ("entries", {"entry_load": 3.0}),
    "short_keys": ("short_keys", {}),
    "short_keys_overflowing": ("short_keys", {"short_load": 12.0, "short_line_cap": 10}),
    "wide_entries": ("wide_entries", {}),
    "full_keys": ("full_keys", {}),
    "full_keys_k31": ("full_keys", {}),
    "full_keys_crowded": ("full_keys", {"full_load": 9.0}),   # lines that fill: keys past their line, windows that walk
}


def _lists(gpu, kind, k, n):
    """device key arrays (d_a, d_b, n_a, n_b) and a closer"""
    from trio_binning_amd._lib import check, lib

    def dalloc(nbytes):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(0, nbytes, C.byref(p)))
        return p.value

    if kind == "uniform":
        d = dalloc(2 * n * 8)
        check(lib.tbk_synth_keys_device(0, 0x5EED0001, 0, 2 * n, k, C.c_void_p(d)))
        return d, d + 8 * n, n, n, d
    cap = int(n * 1.3) + 1024
    d = dalloc(2 * cap * 8)
    got = C.c_uint64()
    snp = 1 / 300
    p_diff = 2 * snp - snp ** 2 * (1 + 1 / 3)
    check(lib.tbk_synth_hap_keys_device(0, 0x5EED0001, int(n / (1 - (1 - p_diff) ** k)), int(snp * (1 << 24)), k, C.c_void_p(d), C.c_void_p(d + 8 * cap), cap, C.byref(got)))
    assert 0.7 * n < got.value <= cap
    return d, d + 8 * cap, got.value, got.value, d


@pytest.mark.parametrize("kind", ["uniform", "haplotypes"])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_sweep_small(gpu, layout, kind):
    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib
    from trio_binning_amd.sweep import full_membership_sweep

    k = 27 if layout in ("wide_entries", "full_keys") else 31 if layout == "full_keys_k31" else 23 if layout == "full_keys_crowded" else 21
    n = 300_000
    d_a, d_b, n_a, n_b, base = _lists(gpu, kind, k, n)
    # hapB's list also gets 2000 of hapA's keys (they must count for hapA: c/kmers.c:291-294)
    ha = np.empty(n_a, dtype=np.uint64)
    hb = np.empty(n_b, dtype=np.uint64)
    check(lib.tbk_memcpy_d2h(0, ha.ctypes.data, C.c_void_p(d_a), n_a * 8))
    check(lib.tbk_memcpy_d2h(0, hb.ctypes.data, C.c_void_p(d_b), n_b * 8))
    check(lib.tbk_device_free(0, C.c_void_p(base)))
    hb = np.concatenate([hb, ha[:2000]])
    name, fields = LAYOUTS[layout]
    with kmers.HashSet.from_keys(ha, k) as a, kmers.HashSet.from_keys(hb, k) as b, kmers.Classifier(a, b, options=kmers.Options.layout(name, **fields)) as cls:
        st = cls.stats()
        assert st["entry_layout"] == layout.startswith(("entries", "wide")) and st["short_keys"] == layout.startswith("short"), st
        assert st["wide_entries"] == (layout == "wide_entries") and (layout != "key_front" or st["front_layout"]) and st["full_keys"] == layout.startswith("full")
        if layout == "full_keys_crowded":
            assert st["keys_past_half"] > 0, st   # keys that left their line
        assert st["shared_keys"] >= 2000
        rec = full_membership_sweep(cls, a, b, a.device_keys, b.device_keys, ha.size, hb.size, k, chunk=1 << 17)
        assert rec["ok"], [r for r in rec["legs"] if not r["ok"]]
        assert len(rec["legs"]) == 7 and rec["shared_keys"] == st["shared_keys"]
        by = {r["leg"]: r for r in rec["legs"]}
        assert by["members_hapA_k_base_reads"]["sum_a"] == ha.size and by["members_hapB_k_base_reads"]["sum_b"] == hb.size - st["shared_keys"]
        # near misses of a clustered list are often members themselves (the neighbouring haplotype's k-mer): the standalone
        # tables say which, and the paired table must agree - so the leg's sums are not all zero there
        if kind == "haplotypes":
            assert by["near_misses_of_hapA_keys"]["sum_a"] + by["near_misses_of_hapA_keys"]["sum_b"] > 0


def test_two_classifiers_with_different_layouts_built_at_the_same_time(gpu):
    """tbk_options instead of the environment: five threads build five classifiers over the same two lists at the same
    time, each with its own pinned layout; every one gets the layout it asked for and answers like the others."""
    import threading

    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib

    k, n = 21, 200_000
    keys = np.empty(2 * n, dtype=np.uint64)
    check(lib.tbk_synth_keys_host(0x5EED0001, 0, 2 * n, k, keys.ctypes.data))
    rng = np.random.default_rng(5)
    reads = []
    for i in range(64):
        r = rng.integers(0, 4, int(rng.integers(30, 4000)))
        key = int(keys[int(rng.integers(0, 2 * n))])
        if r.size > 60:
            r[10:10 + k] = [(key >> (2 * j)) & 3 for j in range(k)]
        reads.append("".join("ACGT"[c] for c in r))
    bases, offsets = kmers.pack_reads(reads)
    wanted = ["keys_front", "keys_whole_lines", "entries", "short_keys", "full_keys"]
    got, errors = {}, []
    with kmers.HashSet.from_keys(keys[:n], k) as a, kmers.HashSet.from_keys(keys[n:], k) as b:
        start = threading.Barrier(len(wanted))

        def build(name):
            try:
                start.wait()
                with kmers.Classifier(a, b, options=kmers.Options.layout(name, memory_budget_bytes=1 << 30)) as cls:
                    got[name] = (cls.stats(), cls.classify_batch(bases, offsets))
            except Exception as exc:  # noqa: BLE001
                errors.append((name, exc))

        threads = [threading.Thread(target=build, args=(name,)) for name in wanted]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors
    assert got["keys_front"][0]["front_layout"] and not got["keys_front"][0]["entry_layout"] and not got["keys_front"][0]["short_keys"]
    assert not got["keys_whole_lines"][0]["front_layout"] and not got["keys_whole_lines"][0]["short_keys"]
    assert got["entries"][0]["entry_layout"] and got["short_keys"][0]["short_keys"] and got["full_keys"][0]["full_keys"]
    assert all(st["table_bytes"] <= 1 << 30 for st, _ in got.values())
    first = got[wanted[0]][1]
    assert first.sum() >= 50 and all(np.array_equal(c, first) for _, c in got.values())


def test_sweep_notices_wrong_answers(gpu):
    """Controls: the same sweep against a table that lacks keys, or holds keys it should not, must report them."""
    from trio_binning_amd import kmers
    from trio_binning_amd._lib import check, lib
    from trio_binning_amd.sweep import _sweep, full_membership_sweep

    k, n = 21, 100_000
    keys = np.empty(3 * n, dtype=np.uint64)
    check(lib.tbk_synth_keys_host(0x5EED0001, 0, 3 * n, k, keys.ctypes.data))
    with kmers.HashSet.from_keys(keys[:n], k) as a, kmers.HashSet.from_keys(keys[n:2 * n], k) as b, \
            kmers.HashSet.from_keys(keys[: n - 7], k) as a_short, kmers.Classifier(a_short, b) as lacking:
        # the classifier was built WITHOUT hapA's last 7 keys: sweeping the full list finds exactly those
        r = _sweep(lacking, a.device_keys, n, k, 1, 1)
        assert r["bad_reads"] == 7 and r["first_bad"] == n - 7 and r["sum_a"] == n - 7
        r = _sweep(lacking, a.device_keys, n, k, 512, 1)
        assert r["bad_reads"] == 1 and r["sum_a"] == n - 7   # all seven sit in the last long read
        # keys expected to be absent that are present
        r = _sweep(lacking, b.device_keys, n, k, 1, 0)
        assert r["bad_reads"] == n and r["sum_b"] == n and r["first_bad"] == 0
        # ... and counted for the wrong list
        r = _sweep(lacking, b.device_keys, n, k, 1, 1)
        assert r["bad_reads"] == n
        # the field check (tbk_classifier_verify; the CLI's TBK_VERIFY_BUILD=1): the classifier lacks seven of `a`'s lines
        v = lacking.verify(a, b)
        assert v == {"lines": 2 * n, "count_a": n - 7, "count_b": n, "bad_lines": 7, "first_bad": n - 7}, v
        assert lacking.verify()["bad_lines"] == 0   # ... and is complete for the lists it was built from
        rec = full_membership_sweep(lacking, a, b, a.device_keys, b.device_keys, n, n, k, uniform_seed=0x5EED0001, chunk=1 << 16)
        assert not rec["ok"]
        assert [x["leg"] for x in rec["legs"] if not x["ok"]] == ["members_hapA_k_base_reads", "members_hapA_long_reads"]
