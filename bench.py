#!/usr/bin/env python3
"""bench.py — Gbases/s classified on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

Workload (N=1 default = BASELINE.json configs[2], the configuration the metric is quoted
on): k = 21, two 300 M-entry synthetic unique-k-mer tables replicated in each GPU's HBM,
synthetic 15 kb reads with planted list k-mers.  A *step* is one pass of the
hot path over one batch of reads that is already resident in HBM when the timed region
starts: zero the counts, run the probe kernel, copy the per-read counts to the host and
take the A/B/U binning decision there (the host part of step i-1 overlaps the kernel of
step i; every step's host part is inside the timed region).  `value` = bases classified by all ranks / wall time
between two barriers.  Reads are sharded across ranks (each rank draws its own reads from
the generator), tables are replicated, there is no collective on the data path; the only
cross-rank traffic is the barrier and the max/sum of the timing (gloo).

The JSON line also carries
  roofline      the probe kernel's algorithmic bytes per launch / its HIP-event-timed
                average duration, against the 8 TB/s HBM peak;
  cpu_baseline  the oracle (faithful CPU restatement of c/kmers.c) timed on this box's
                host cores on a bounded sample of the same reads and tables, with a
                count-for-count parity check of the GPU result on that sample.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEY_SEED = 0x5EED0001
READ_SEED = 0x5EED0002
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured stream)


def parse():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--kmers-per-list", type=int, default=300_000_000)
    ap.add_argument("--read-len", type=int, default=15_000)
    ap.add_argument("--reads-per-step", type=int, default=65_536)
    ap.add_argument("--resident-batches", type=int, default=4, help="distinct read batches kept in HBM and cycled")
    ap.add_argument("--lists", choices=["uniform", "haplotypes"], default="uniform",
                    help="uniform: BASELINE.json's synthetic lists (distinct uniform random k-mers, reads with planted list "
                         "k-mers); haplotypes: lists shaped like real find-unique-kmers output (two haplotypes of a random "
                         "genome differing by SNPs; reads drawn from them with errors)")
    ap.add_argument("--snp-rate", type=float, default=1 / 500, help="haplotypes: SNPs per base of each haplotype")
    ap.add_argument("--error-rate", type=float, default=0.002, help="haplotypes: substitution errors per read base")
    ap.add_argument("--plant-major", type=int, default=30)
    ap.add_argument("--plant-minor", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time per cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calibrate", action="store_true", help="also run the random-line gather calibration")
    ap.add_argument("--share-device", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses device 0 (numbers are not a scaling result)")
    return ap.parse_args()


def spawn_ranks(args):
    """`--gpus N` without a launcher: start N ranks as child processes (never exec)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Dist:
    """Barrier + max/sum over ranks.  gloo on CPU tensors: the data path has no collective,
    so nothing here touches the GPU."""

    def __init__(self, world):
        self.world = world
        self.dist = None
        if world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", init_method="env://")
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def reduce(self, value, op):
        if not self.dist:
            return value
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t[0])

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def shard_plan(total_units, rank, world):
    """Contiguous shard [lo, hi) of `total_units` for `rank` (weak scaling uses it with
    total = per_rank * world, strong scaling with a fixed total)."""
    lo = total_units * rank // world
    hi = total_units * (rank + 1) // world
    return lo, hi


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit(spawn_ranks(args))

    import fcntl

    import numpy as np

    import __graft_entry__ as entry

    # every rank makes sure the library is built (make is a no-op when it is), one at a time
    if os.environ.get("TBK_SKIP_BUILD") != "1":  # skipped under rocprofv3 (no child processes there)
        with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            entry.build()
            fcntl.flock(lock, fcntl.LOCK_UN)
    # Load the HIP library and initialise its runtime BEFORE torch is imported: the torch wheel
    # bundles its own copy of the HIP runtime and whichever copy is loaded first owns the process.
    from trio_binning_amd import _lib, kmers
    from trio_binning_amd._lib import check, lib

    n_dev = _lib.device_count()
    dist = Dist(world)
    dist.barrier()

    dev = 0 if args.share_device else local_rank
    if n_dev <= dev:
        raise SystemExit(f"rank {rank}: HIP device {dev} not visible ({n_dev} devices)")
    k, n_list, L, R = args.k, args.kmers_per_list, args.read_len, args.reads_per_step

    def dalloc(nbytes):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, nbytes, C.byref(p)))
        return p.value

    t_setup = time.time()
    # ---- tables: 2 x n_list distinct canonical k-mers, generated and inserted on the GPU ----
    hap = args.lists == "haplotypes"
    if hap:
        # genome long enough for ~n_list windows that cover a position where the haplotypes differ
        snp24 = max(1, int(round(args.snp_rate * (1 << 24))))
        err24 = int(round(args.error_rate * (1 << 24)))
        p_diff = 2 * args.snp_rate - args.snp_rate ** 2 * (1 + 1 / 3)
        genome_len = int(n_list / (1 - (1 - p_diff) ** k))
        cap = int(n_list * 1.05) + 1024
        d_keys = dalloc(2 * cap * 8)
        n_got = C.c_uint64()
        check(lib.tbk_synth_hap_keys_device(dev, KEY_SEED, genome_len, snp24, k, C.c_void_p(d_keys), C.c_void_p(d_keys + cap * 8),
                                            cap, C.byref(n_got)))
        if n_got.value > cap:
            raise SystemExit(f"haplotype lists: {n_got.value} keys exceed the capacity {cap}")
        n_list, key_stride = n_got.value, cap
    else:
        d_keys = dalloc(2 * n_list * 8)
        check(lib.tbk_synth_keys_device(dev, KEY_SEED, 0, 2 * n_list, k, C.c_void_p(d_keys)))
        key_stride = n_list
    t0 = time.time()
    hap_a = kmers.HashSet.from_device_keys(d_keys, n_list, k, device=dev)
    hap_b = kmers.HashSet.from_device_keys(d_keys + key_stride * 8, n_list, k, device=dev)
    cls = kmers.Classifier(hap_a, hap_b)  # hashes both lists into the paired table in HBM
    check(lib.tbk_device_sync(dev))
    t_build = time.time() - t0
    stats = cls.stats()
    if not hap:
        assert stats["distinct_a"] == n_list and stats["distinct_b"] == n_list, stats

    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    h_keys = None
    if want_cpu:
        h_keys = np.empty(2 * n_list, dtype=np.uint64)
        check(lib.tbk_memcpy_d2h(dev, h_keys.ctypes.data, C.c_void_p(d_keys), n_list * 8))
        check(lib.tbk_memcpy_d2h(dev, h_keys.ctypes.data + n_list * 8, C.c_void_p(d_keys + key_stride * 8), n_list * 8))
    check(lib.tbk_device_free(dev, C.c_void_p(d_keys)))

    # ---- reads: `resident_batches` batches in HBM; each rank draws its own reads --------------
    nb = max(1, args.resident_batches)
    total = R * L
    batches = []
    for b in range(nb):
        first_read = (rank * nb + b) * R  # disjoint read indices per rank: sharded, weak scaling
        d_bases = dalloc((total + 15) // 16 * 16 + 16)
        d_offs = dalloc((R + 1) * 8)
        d_counts = dalloc(R * 2 * 4)
        if hap:
            check(lib.tbk_synth_hap_reads_device(dev, KEY_SEED, genome_len, snp24, READ_SEED, first_read, R, L, err24,
                                                 C.c_void_p(d_bases), C.c_void_p(d_offs)))
        else:
            check(lib.tbk_synth_reads_device(dev, READ_SEED, first_read, R, L, KEY_SEED, n_list, n_list, k,
                                             args.plant_major, args.plant_minor, C.c_void_p(d_bases), C.c_void_p(d_offs)))
        batches.append((d_bases, d_offs, d_counts))
    t_setup = time.time() - t_setup

    depth = cls.depth
    counts_ring = [kmers.pinned_empty((R, 2), np.int32) for _ in range(depth)]
    num_a, num_b = hap_a.num_kmers, hap_b.num_kmers
    bins_total = {"A": 0, "B": 0, "U": 0}

    def finish(ticket, slot, tally):
        """Host side of a step: wait for its counts, take the A/B/U decision."""
        cls.wait(ticket)
        _, _, bins = kmers.score_and_bin(counts_ring[slot], num_a, num_b)
        if tally:
            for name, ch in (("A", b"A"), ("B", b"B"), ("U", b"U")):
                bins_total[name] += bins.count(ch)

    def run(n_steps, tally):
        """n_steps steps, pipelined: the host finishes step i-1 while the GPU probes step i."""
        pending = []
        for i in range(n_steps):
            d_bases, d_offs, _ = batches[i % nb]
            slot = i % depth
            if len(pending) == depth - 1 + (depth == 1):
                finish(*pending.pop(0), tally)
            pending.append((cls.submit_device(d_bases, d_offs, R, total, counts_ring[slot]), slot))
        while pending:
            finish(*pending.pop(0), tally)

    run(args.warmup, False)
    cls.kernel_timing(True)
    check(lib.tbk_device_sync(dev))
    dist.barrier()
    t0 = time.perf_counter()
    run(args.steps, True)
    check(lib.tbk_device_sync(dev))
    dist.barrier()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = cls.kernel_timing_read()
    cls.kernel_timing(False)

    elapsed_max = dist.reduce(elapsed, "MAX")
    bases_all = dist.reduce(args.steps * total, "SUM")
    value = bases_all / elapsed_max / 1e9

    if not stats["minimizer_w"]:
        bucket_select = "plain hash"
    elif stats["sampling_t"]:
        bucket_select = "mod-sampling w=%d m=%d t=%d" % (stats["minimizer_w"], stats["minimizer_m"], stats["sampling_t"])
    else:
        bucket_select = "minimizer w=%d m=%d" % (stats["minimizer_w"], stats["minimizer_m"])
    # ---- roofline of the probe kernel (rank 0's device) -----------------------------------------
    # algorithmic bytes per window: 1 read byte + 8 B for the hapA slot + 8 B for the
    # hapB slot when hapA missed.  Per launch: windows = R * (L - k + 1).
    counts = counts_ring[0]
    dbg_read = getattr(lib, "tbk_debug_counters", None) if hasattr(lib, "tbk_debug_counters") else None
    if dbg_read is not None:  # debug build (-DTBK_COUNTERS): event counts of one launch, to stderr
        buf = (C.c_ulonglong * 8)()
        dbg_read(buf, 1)
    finish(cls.submit_device(batches[0][0], batches[0][1], R, total, counts), 0, False)
    if dbg_read is not None:
        dbg_read(buf, 1)
        w_ = R * max(1, L - k + 1)
        print("tbk-counters", json.dumps({"careful_jstep_frac": round(buf[1] / max(buf[0], 1), 4), "careful_substeps_per_jstep": round(buf[2] / max(buf[0], 1), 4),
                                         "walks_per_window": round(buf[3] / w_, 6), "lines_per_window": round(buf[4] / 4 / w_, 4)}), file=sys.stderr)
    hits_a = int(counts[:, 0].sum())
    windows = R * max(0, L - k + 1)
    alg_bytes = windows * 9 + (windows - hits_a) * 8
    avg_kernel_s = kernel_ms / max(1, launches) * 1e-3
    achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    traffic = None
    for tname in ("pmc_traffic.json", "pmc_traffic_haplotypes.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if not os.path.isfile(tfile):
            continue
        try:
            t = json.load(open(tfile))
            same = (t.get("reads_per_step") == R and t.get("read_len") == L and t.get("kmers_per_list") == n_list
                    and t.get("k") == k and t.get("bucket_select") == bucket_select and t.get("lists", "uniform") == args.lists
                    and abs(t.get("table_load", 0) - n_list / (stats["n_buckets"] * 8)) < 1e-3)
            if same:  # measured in a separate rocprofv3 --pmc pass on this exact configuration
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            pass
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
        "kernel": "tbk_probe_kernel", "kernel_ms_avg": round(avg_kernel_s * 1e3, 4), "launches": int(launches),
        "alg_bytes_per_launch": int(alg_bytes), "alg_bytes_per_window": round(alg_bytes / max(1, windows), 3),
        "kernel_only_gbases_per_s": round(total / avg_kernel_s / 1e9, 2) if avg_kernel_s > 0 else None,
    }
    if traffic is not None and avg_kernel_s > 0:
        # where the kernel sits against what the memory system can actually deliver: PMC-measured
        # bytes per launch over this run's kernel time, and 128-byte lines per second against the
        # random-line ceiling measured by tools/calib_footprint.py (profiles/calibration.json)
        roofline["traffic_GBps"] = round(traffic / avg_kernel_s / 1e9, 1)
        roofline["traffic_frac_of_peak"] = round(traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBPS, 4)
        cfile = os.path.join(ROOT, "profiles", "calibration.json")
        if os.path.isfile(cfile):
            try:
                cal = json.load(open(cfile))
                # the calibration row whose footprint is closest to this table's
                rows = cal["random_lines_Glines_per_s"]
                key = min(rows, key=lambda name: abs(float(name[:-2]) - stats["table_bytes"] / 1e9))
                ceiling = rows[key]["line128"]
                roofline["random_line_ceiling_footprint"] = key
                roofline["random_lines_Gps"] = round(traffic / 128 / avg_kernel_s / 1e9, 2)
                roofline["random_line_ceiling_Gps"] = ceiling
                roofline["random_line_frac"] = round(traffic / 128 / avg_kernel_s / 1e9 / ceiling, 3)
                roofline["traffic_frac_of_measured_stream"] = round(traffic / avg_kernel_s / 1e9 / cal["guide_stream_GBps"], 3)
            except Exception:
                pass

    if want_cpu and L > 2_000_000:
        # the CPU sample works in whole reads; one read of this length is minutes of oracle time
        print(f"bench: reads of {L} bases are too long for a bounded CPU sample: cpu_baseline skipped", file=sys.stderr)
        want_cpu = False
    out = {
        "metric": "Gbases/sec classified (k=21, 2x300M k-mer tables)", "value": round(value, 3), "unit": "Gbases/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64",
        "data": "synthetic" if not hap else f"synthetic haplotypes (SNP rate {args.snp_rate:g}, read error rate {args.error_rate:g})",
        "config": {
            "workload": f"BASELINE configs[2] shape: {L} b synthetic reads, 2x{n_list} unique {k}-mers replicated per GPU, "
                        f"{R} reads ({total / 1e9:.3f} Gbases) per step per GPU resident in HBM, reads sharded over ranks",
            "k": k, "kmers_per_list": n_list, "read_len": L, "reads_per_step": R, "resident_batches": nb,
            "table_bytes_per_gpu": stats["table_bytes"], "table_load": round(n_list / (stats["n_buckets"] * 8), 4),
            "bucket_select": bucket_select, "lists": args.lists,
            "parallelism": f"read-sharded x{world}, tables replicated, no data-path collective",
        },
        "roofline": roofline,
        "bins": bins_total, "setup_s": round(t_setup, 2), "table_build_s": round(t_build, 2),
        "device": _lib.device_name(dev),
    }

    if args.calibrate and rank == 0:
        out["calibration"] = calibrate(lib, check, dev, stats["table_bytes"])

    # ---- CPU baseline + parity on a bounded sample (rank 0, N=1 only) -----------------------------
    if want_cpu:
        out["cpu_baseline"], out["parity"] = cpu_baseline(args, np, lib, check, dev, batches[0], counts, h_keys, n_list, k, L, R)

    cls.close()
    dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.close()


def calibrate(lib, check, dev, footprint):
    res = {}
    bps = C.c_double()
    check(lib.tbk_calib_stream(dev, min(footprint, 8 << 30), 5, C.byref(bps)))
    res["stream_GBps"] = round(bps.value / 1e9, 1)
    for line, lpl in ((64, 4), (64, 1), (128, 8), (128, 1)):
        for inf in (1, 2, 4, 8):
            lps, ms = C.c_double(), C.c_double()
            check(lib.tbk_calib_gather(dev, footprint, line, lpl, inf, 1 << 28, 3, C.byref(lps), C.byref(ms)))
            res[f"gather_line{line}_lanes{lpl}_inflight{inf}"] = {
                "Glines_per_s": round(lps.value / 1e9, 2), "GBps": round(lps.value * line / 1e9, 1)}
    return res


def cpu_baseline(args, np, lib, check, dev, batch0, gpu_counts_batch0, h_keys, n_list, k, L, R):
    """The oracle on this box's host cores, same tables, a prefix of batch 0's reads."""
    import oracle

    orc = oracle.load()
    # CPUs this process may actually use: hardware threads cut down to the affinity mask and the
    # cgroup CPU quota (the GPU boxes show 256 hardware threads and grant 16 CPUs' worth of time)
    cores = int(lib.tbk_host_threads())
    hw_threads = os.cpu_count() or 1
    t0 = time.time()
    oa = orc.table_from_keys(h_keys[:n_list], k, threads=cores)
    ob = orc.table_from_keys(h_keys[n_list:], k, threads=cores)
    t_tables = time.time() - t0
    sample_reads = min(R, 4096)
    h_bases = np.empty(sample_reads * L, dtype=np.uint8)
    check(lib.tbk_memcpy_d2h(dev, h_bases.ctypes.data, C.c_void_p(batch0[0]), h_bases.nbytes))
    offs = (np.arange(sample_reads + 1, dtype=np.uint64) * np.uint64(L))

    def timed(n_reads, threads):
        t = time.perf_counter()
        c = orc.count_batch(h_bases[: n_reads * L], offs[: n_reads + 1], oa, ob, threads=threads)
        return time.perf_counter() - t, c

    # 1 thread = what the reference does.  Calibrate on a few reads, then ~cpu_seconds worth.
    probe_n = min(sample_reads, 16)
    dt, _ = timed(probe_n, 1)
    n1 = int(max(probe_n, min(sample_reads, args.cpu_seconds / max(dt / probe_n, 1e-9))))
    dt1, c1 = timed(n1, 1)
    rate1 = n1 * L / dt1 / 1e9
    # all host cores, reads sharded over threads sharing the read-only tables
    nall = int(max(n1, min(sample_reads, n1 * cores * 0.7)))
    dtn, cn = timed(nall, cores)
    raten = nall * L / dtn / 1e9
    # fairness datum (SURVEY 8d CPU-opt): rolling k-mers + all threads on the same tables; not the
    # reference's algorithm
    t = time.perf_counter()
    cf = orc.count_batch_fast(h_bases[: sample_reads * L], offs[: sample_reads + 1], oa, ob, threads=cores)
    dtf = time.perf_counter() - t
    ratef = sample_reads * L / dtf / 1e9
    # parity: GPU counts of batch 0 (from the roofline step) vs the oracle on the sample
    g = gpu_counts_batch0[:nall]
    equal = (bool(np.array_equal(g, cn)) and bool(np.array_equal(g[:n1], c1))
             and bool(np.array_equal(gpu_counts_batch0[:sample_reads], cf)))
    parity = {"reads_checked": int(nall), "bases_checked": int(nall * L), "gpu_equals_cpu": equal,
              "count_checksum": [int(cn[:, 0].sum()), int(cn[:, 1].sum())]}
    if not equal:
        bad = np.nonzero((g != cn).any(axis=1))[0][:5]
        parity["first_mismatches"] = [[int(i), g[i].tolist(), cn[i].tolist()] for i in bad]
    base = {
        "value": round(rate1, 6), "unit": "Gbases/s", "cores": 1, "kind": "port", "host_hardware_threads": hw_threads, "host_usable_cpus": cores,
        "sample": f"oracle (faithful restatement of c/kmers.c: 2 linear-probe tables at load 0.75, non-rolling encode) "
                  f"on the first {n1} reads ({n1 * L / 1e6:.1f} Mbases) of batch 0, same 2x{n_list} {k}-mer tables, {dt1:.1f} s",
        "all_cores": {"value": round(raten, 6), "unit": "Gbases/s", "cores": cores,
                      "sample": f"first {nall} reads ({nall * L / 1e6:.1f} Mbases), reads sharded over {cores} threads, {dtn:.1f} s"},
        "optimised_rolling_all_cores": {"value": round(ratef, 6), "unit": "Gbases/s", "cores": cores,
                                        "sample": f"fairness datum, not the reference's algorithm: rolling canonical k-mers, "
                                                  f"{sample_reads} reads ({sample_reads * L / 1e6:.1f} Mbases) over {cores} threads, {dtf:.1f} s"},
        "table_build_s": round(t_tables, 1),
    }
    return base, parity


if __name__ == "__main__":
    main()
