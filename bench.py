#!/usr/bin/env python3
"""bench.py — Gbases/s classified on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

Workload (N=1 default = BASELINE.json configs[2], the configuration the metric is quoted
on): k = 21, two 300 M-entry synthetic unique-k-mer tables replicated in each GPU's HBM,
synthetic 15 kb reads with planted list k-mers.

A *step* is one pass of the classify stage over one batch of reads: the batch lies in
pinned host memory in the form the product's reader hands batches over (its bases in the packed
transfer format, tbk_fastx_set_packing), goes through the library's pipeline - queue, feeder thread,
H2D on the side stream overlapped with the previous batch's kernel, probe kernels, D2H of the counts
- and the host takes the A/B/U binning decision (the host part of step i-1 overlaps the device part of
step i; every step's host part is inside the timed region).  `value` = bases classified by all ranks /
wall time of K steps between two barriers (max over ranks): the host-fed, PCIe-inclusive rate.
`kernel_resident` beside it is the same K steps with the batches already in HBM (no H2D): what the
kernels alone sustain.  (`--timed-path resident` makes that one the `value`, as rounds 1-2 reported it.)
A region of K steps that lasts less than --min-timed-s is repeated until that much time has been
measured, and the median region is reported (`timed_regions` says how many).

Scaling.  Default "weak": every rank classifies its own reads (the generator is indexed by
read number, ranks take disjoint ranges), tables replicated, no collective on the data path; every rank
runs the host-fed stage at the same time (what N GPUs of one node share is the host: DRAM, PCIe root
complexes, cores).  `--scaling strong` is BASELINE configs[3]: ONE fixed read set (--strong-reads reads,
default the 90 Gbp of configs[3]) split over the ranks by read index; a step is one pass of a rank over
its whole shard, `value` = the set's bases / the slowest rank's time.

Ranks are started by the driver's launcher (`python -m torch.distributed.run ... bench.py`, which
only sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT) or, without one, by this script itself.
Nothing here imports torch: the barrier and the max/sum of the timing scalars go through files in
a run-private directory (the ranks of one node share /tmp); `TBK_BENCH_DIST=gloo` selects a
torch.distributed gloo group instead (tests cover both).

The JSON line also carries, at every N,
  roofline      the dominant kernel's (the single-read probe kernel's) algorithmic bytes per launch / its
                HIP-event-timed average duration inside the timed region, against the 8 TB/s HBM peak.  `frac` is
                SURVEY §8d's reading for this design - one paired table probed once per window, P = 1: 9 bytes
                per window; the two-probe reading (P = 2: 17 bytes) is printed beside it.  `traffic` (N = 1) is measured
                in the run: when the timed legs are over, rank 0 repeats four steps of the workload as a child process under
                `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (live_traffic; --live-pmc off skips it); where that pass cannot
                be made, the round's PMC record under profiles/ is replayed - only when it was taken on this tree's kernels
                (sha256 of their machine code) - and `traffic_source` says which of the two it was;
  parity        every rank classifies the same fixed reads (read 0 .. 4095 of the generator) through the
                host-fed path; the count checksums must agree across ranks, and rank 0 checks the counts
                read for read against the oracle;
  cpu_baseline  the oracle (faithful CPU restatement of c/kmers.c) timed on rank 0's host cores on a
                bounded sample of the same reads and tables;
  devices       each rank's device (PCI bus id, uuid), so that N distinct GPUs are provable;
and at N = 1 `pipeline_variants`: the same stage fed with ASCII batches (packed by the feeder thread
before the copy) and with ASCII over PCIe (the kernel packs), and `realistic_lists`: the same bench on lists
shaped like real find-unique-kmers output (`--lists haplotypes`), in the same run.

`--path count` benches the k-mer counting kernel of the find-unique-kmers step instead
(SURVEY §8f N4): Gbases/s counted, atomic adds per second against the chip's measured ceiling,
a CPU datum and a histogram parity check.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEY_SEED = 0x5EED0001
READ_SEED = 0x5EED0002
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured stream)
BASELINE_K, BASELINE_KEYS = 21, 300_000_000


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--path", choices=["classify", "count"], default="classify")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--strong-reads", type=int, default=6_000_000, help="strong scaling: reads of the fixed set (configs[3]: 6 M x 15 kb = 90 Gbp)")
    ap.add_argument("--min-timed-s", type=float, default=10.0,
                    help="repeat the K-step region until this much time has been measured (10 s: a sustained-clock number the driver's "
                         "utilisation sampler can see)")
    ap.add_argument("--no-realistic", action="store_true",
                    help="N = 1: skip the `realistic_lists` sub-record (the same bench on haplotype-shaped lists, a second table build and a few seconds of timing)")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="N = 1: skip the `strong_90gbp` sub-record (BASELINE configs[2]'s literal set: one pass per step over --strong-reads distinct reads)")
    ap.add_argument("--strong-leg-timed-s", type=float, default=1.2, help="timed seconds of the strong_90gbp sub-record (whole passes of 0.43 s)")
    ap.add_argument("--realistic-timed-s", type=float, default=3.0, help="timed seconds per leg of the realistic_lists sub-record")
    ap.add_argument("--k", type=int, default=BASELINE_K)
    ap.add_argument("--kmers-per-list", type=int, default=BASELINE_KEYS)
    ap.add_argument("--read-len", type=int, default=15_000)
    ap.add_argument("--reads-per-step", type=int, default=262_144, help="reads of one batch = one step (3.9 Gbases at 15 kb)")
    ap.add_argument("--read-lengths", choices=["fixed", "lognormal"], default="fixed",
                    help="lognormal: BASELINE configs[4] as SURVEY 8d writes it - log-normal lengths with N50 --n50 (a tail past 1 Mb, "
                         "a floor of short reads); --read-len is then only the figure --reads-per-step is sized by")
    ap.add_argument("--n50", type=float, default=100_000.0)
    ap.add_argument("--sigma", type=float, default=0.9, help="lognormal: sigma of ln(length)")
    ap.add_argument("--short-fraction", type=float, default=0.05, help="lognormal: share of the reads that are debris of 1 b .. 5 kb")
    ap.add_argument("--max-read-len", type=int, default=4_000_000)
    ap.add_argument("--no-sweep", dest="sweep", action="store_false",
                    help="skip the full-membership sweep of the built table (every list key, as many non-members, every key's near "
                         "miss - trio_binning_amd/sweep.py; under a second at 2 x 3e8 keys, recorded under `sweep`)")
    ap.add_argument("--resident-batches", type=int, default=2, help="weak scaling: distinct read batches kept in HBM and cycled")
    ap.add_argument("--lists", choices=["uniform", "haplotypes"], default="uniform",
                    help="uniform: BASELINE.json's synthetic lists (distinct uniform random k-mers, reads with planted list "
                         "k-mers); haplotypes: lists shaped like real find-unique-kmers output (two haplotypes of a random "
                         "genome differing by SNPs; reads drawn from them with errors)")
    ap.add_argument("--snp-rate", type=float, default=1 / 500, help="haplotypes: SNPs per base of each haplotype")
    ap.add_argument("--repeat-fraction", type=float, default=0.0,
                    help="haplotypes: this fraction of the genome's 8-kb blocks are copies of one of 16 family sequences, "
                         "2 %% diverged (young interspersed repeats: the lists' crowded buckets)")
    ap.add_argument("--error-rate", type=float, default=0.002, help="haplotypes: substitution errors per read base")
    ap.add_argument("--plant-major", type=int, default=30)
    ap.add_argument("--plant-minor", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time per cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streaming", action="store_true", help="skip the pipeline variants (ASCII batches in; ASCII over PCIe)")
    ap.add_argument("--timed-path", choices=["host_fed", "resident"], default="host_fed",
                    help="what `value` times: the host-fed classify stage (pinned packed batches -> H2D -> kernels -> D2H -> binning), "
                         "or the same steps with the batches resident in HBM")
    ap.add_argument("--rings", type=int, default=1,
                    help="feeder threads + stream rings (+ table replicas) per rank, all on the rank's device: 1 is the product's "
                         "shape per GPU; more exercise the pipeline's dealing on a one-GPU box")
    ap.add_argument("--parity-reads", type=int, default=4096, help="reads of the generator's start every rank classifies for the parity check")
    ap.add_argument("--stream-batch-reads", type=int, default=65_536, help="reads per host batch of the streaming leg")
    ap.add_argument("--stream-seconds", type=float, default=2.0)
    ap.add_argument("--calibrate", action="store_true", help="also run the random-line gather calibration")
    ap.add_argument("--live-pmc", choices=["auto", "on", "off"], default="auto",
                    help="roofline.traffic from a counter pass of THIS run: when the timed legs are done, rank 0 starts `rocprofv3 --kernel-trace --pmc "
                         "FETCH_SIZE -- python3 bench.py <the same workload, 4 steps>` as a child process and prices the single-read kernel's HBM "
                         "bytes from its counter file.  auto: N = 1, fixed read lengths, weak scaling, rocprofv3 on PATH, not itself under a profiler; "
                         "off (or a failed pass): the record of profiles/pmc_traffic*.json is replayed when it was taken on these very kernels")
    ap.add_argument("--share-device", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses device 0 (numbers are not a scaling result)")
    # --path count
    ap.add_argument("--count-genome", type=int, default=200_000_000)
    ap.add_argument("--count-read-len", type=int, default=150)
    ap.add_argument("--count-batch-bases", type=int, default=1_000_000_000)
    return ap.parse_args(argv)


# ---- ranks ------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """`--gpus N` without a launcher: start N ranks as child processes (never exec) and hand them a
    private rendezvous directory."""
    import tempfile

    rdv = tempfile.mkdtemp(prefix="tbk_bench_rdv_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), TBK_BENCH_RDV=rdv)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    try:
        for f in os.listdir(rdv):
            os.unlink(os.path.join(rdv, f))
        os.rmdir(rdv)
    except OSError:
        pass
    return rc


class Dist:
    """Barrier + max/sum over the ranks of one node.  The data path has no collective: ranks only
    meet to start and stop the clock together and to combine their timing scalars.

    backend "file" (default): every rank drops `<seq>.<rank>` holding its value into a directory
    the ranks share and polls until all `world` files of that round are there (an all-gather, from
    which barrier, max and sum follow).  The directory is TBK_BENCH_RDV when this script spawned the
    ranks, else derived from what a launcher gives all its workers alike (MASTER_PORT, the
    launcher's pid).  backend "gloo": torch.distributed on CPU tensors, kept for comparison."""

    def __init__(self, world, rank=None, backend=None, rdv=None, timeout_s=1800.0):
        self.world = world
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.backend = backend or os.environ.get("TBK_BENCH_DIST", "file")
        self.seq = 0
        self.timeout_s = timeout_s
        self.dist = None
        self.dir = None
        if world <= 1:
            return
        if self.backend == "gloo":
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", init_method="env://")
            self.dist = dist
            return
        self.dir = rdv or os.environ.get("TBK_BENCH_RDV") or os.path.join(
            "/tmp", "tbk_bench_rdv_%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid()))
        os.makedirs(self.dir, exist_ok=True)

    def gather_obj(self, obj=None):
        """Every rank's (JSON-able) object, in rank order (a barrier as a side effect)."""
        if self.world <= 1:
            return [obj]
        if self.dist:
            out = [None] * self.world
            self.dist.all_gather_object(out, obj)
            return out
        self.seq += 1
        mine = os.path.join(self.dir, "%d.%d" % (self.seq, self.rank))
        with open(mine + ".tmp", "w") as fh:
            json.dump(obj, fh)
        os.rename(mine + ".tmp", mine)  # atomic: a reader sees the whole value or no file
        out, have, deadline = [None] * self.world, [False] * self.world, time.monotonic() + self.timeout_s
        while True:
            for r in range(self.world):
                if not have[r]:
                    try:
                        with open(os.path.join(self.dir, "%d.%d" % (self.seq, r))) as fh:
                            out[r] = json.load(fh)
                        have[r] = True
                    except (OSError, ValueError):
                        pass
            if all(have):
                break
            if time.monotonic() > deadline:
                raise TimeoutError("bench rendezvous: ranks %s never reached round %d" % ([r for r, h in enumerate(have) if not h], self.seq))
            time.sleep(0.0002)
        # a rank may remove its own file of the round before last: everyone has passed that round
        old = os.path.join(self.dir, "%d.%d" % (self.seq - 2, self.rank))
        if self.seq > 2 and os.path.exists(old):
            os.unlink(old)
        return out

    def gather(self, value=0.0):
        """Every rank's value, in rank order (a barrier as a side effect)."""
        return [float(v) for v in self.gather_obj(float(value))]

    def barrier(self):
        self.gather(0.0)

    def reduce(self, value, op):
        vals = self.gather(value)
        return max(vals) if op == "MAX" else min(vals) if op == "MIN" else sum(vals)

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()
        elif self.dir:
            # A rank's files may only go once nobody can still be polling for them: after the last
            # round every other rank says "bye" (it has read all it ever will), and rank 0, having
            # seen every bye, removes the directory.
            self.barrier()
            if self.rank != 0:
                open(os.path.join(self.dir, "bye.%d" % self.rank), "w").close()
                return
            deadline = time.monotonic() + 60.0
            while not all(os.path.exists(os.path.join(self.dir, "bye.%d" % r)) for r in range(1, self.world)) and time.monotonic() < deadline:
                time.sleep(0.001)
            try:
                for f in os.listdir(self.dir):
                    os.unlink(os.path.join(self.dir, f))
                os.rmdir(self.dir)
            except OSError:
                pass


def shard_plan(total_units, rank, world):
    """Contiguous shard [lo, hi) of `total_units` for `rank` (weak scaling uses it with
    total = per_rank * world, strong scaling with a fixed total)."""
    lo = total_units * rank // world
    hi = total_units * (rank + 1) // world
    return lo, hi


def metric_label(k, n_list):
    """BASELINE.json's metric string when the configuration is BASELINE's, else one that says what ran."""
    if k == BASELINE_K and n_list == BASELINE_KEYS:
        return "Gbases/sec classified (k=21, 2x300M k-mer tables)"
    return "Gbases/sec classified (k=%d, 2x%s k-mer tables)" % (k, ("%dM" % round(n_list / 1e6)) if n_list >= 1e6 else str(n_list))


def timed_regions(run_region, dist, min_timed_s, max_regions=64):
    """Call run_region() (K steps between two barriers; returns this rank's seconds) until the
    regions add up to min_timed_s.  Every rank sees the same max-over-ranks times, so all agree when
    to stop.  Returns the list of max-over-ranks region times."""
    times = []
    while True:
        times.append(dist.reduce(run_region(), "MAX"))
        if sum(times) >= min_timed_s or len(times) >= max_regions:
            return times


# ---- classify ---------------------------------------------------------------------------------------
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
    from trio_binning_amd import _lib, kmers
    from trio_binning_amd._lib import check, lib

    n_dev = _lib.device_count()
    dist = Dist(world, rank)
    dist.barrier()
    dev = 0 if args.share_device else local_rank
    if n_dev <= dev:
        raise SystemExit(f"rank {rank}: HIP device {dev} not visible ({n_dev} devices)")
    # one process per GPU: this rank's threads (the pipeline's feeder, the packers, everything it starts from here on) run on
    # the CPUs of the socket its GPU hangs off, and its share of the node's CPUs is 1 / LOCAL_WORLD_SIZE (tbk_host_threads)
    numa_node, numa_cpus = C.c_int(-1), C.c_int(0)
    check(lib.tbk_numa_bind_to_device(dev, C.byref(numa_node), C.byref(numa_cpus)))
    placement = {"numa_node": numa_node.value, "cpus_bound_to": numa_cpus.value, "host_threads": int(lib.tbk_host_threads())}
    if args.path == "count":
        out = bench_count(args, np, kmers, lib, check, dev, dist, world, rank)
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.close()
        return
    out = run_classify(args, np, kmers, lib, check, _lib, dev, dist, world, rank, placement)
    if rank == 0 and world == 1:
        note = live_traffic(args, out)
        if note:
            out["roofline"]["live_counter_pass"] = note
    out["roofline"].pop("_pricing", None)
    if rank == 0 and world == 1 and not args.no_realistic and args.lists == "uniform" and args.scaling == "weak":
        out["realistic_lists"] = realistic_lists(args, np, kmers, lib, check, _lib, dev, dist, placement)
    if rank == 0 and world == 1 and not args.no_strong_leg and args.lists == "uniform" and args.scaling == "weak" and args.read_lengths == "fixed":
        out["strong_90gbp"] = strong_leg(args, np, kmers, lib, check, _lib, dev, dist, placement, out)
    dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.close()


def strong_leg(args, np, kmers, lib, check, _lib, dev, dist, placement, headline):
    """BASELINE configs[2] to the letter, beside the headline's two cycled batches: ONE pass per step over the whole 90 Gbp set
    (--strong-reads reads = 23 distinct host-fed batches at the default sizes), the same tables, the same host-fed stage
    (`--scaling strong` at N = 1).  Its own parity (the generator's first reads against the oracle) and the bins of the whole set."""
    import copy

    a = copy.copy(args)
    a.scaling = "strong"
    a.steps, a.warmup = 1, 1
    a.min_timed_s = args.strong_leg_timed_s
    a.no_streaming = True
    a.calibrate = False
    a.sweep = False
    a.cpu_seconds = min(args.cpu_seconds, 1.0)
    a.parity_reads = min(args.parity_reads, 512)
    sub = run_classify(a, np, kmers, lib, check, _lib, dev, dist, 1, 0, placement)
    r = sub["roofline"]
    return {
        "workload": sub["config"]["workload"], "value": sub["value"], "unit": sub["unit"], "ms_per_pass": sub["ms_per_step"],
        "reads": a.strong_reads, "bases": sub["config"]["bases_per_step_per_rank"], "distinct_batches": sub["config"]["distinct_batches"],
        "passes_timed": sub["timed_regions"], "timed_total_s": sub["timed_total_s"], "region_s_min_median_max": sub["region_s_min_median_max"],
        "kernel_ms_avg": r["kernel_ms_avg"], "whole_probe_ms_avg": r["whole_probe_ms_avg"], "frac": r["frac"],
        "vs_value": round(sub["value"] / headline["value"], 4) if headline.get("value") else None,
        "bins_of_one_pass": {name: n // max(1, sub["timed_regions"]) for name, n in sub["bins"].items()}, "parity": sub["parity"], "setup_s": sub["setup_s"],
        "note": "the headline cycles config.distinct_batches batches per rank; this leg is one pass per step over the literal set - every batch distinct, host-fed",
    }


def realistic_lists(args, np, kmers, lib, check, _lib, dev, dist, placement):
    """The same bench on lists shaped like real find-unique-kmers output (find_unique_kmers.py:200-233: the k overlapping
    k-mers around every variant, in both lists at once) - `--lists haplotypes` - as a sub-record of the default N = 1
    line: value, resident rate, kernel time, both roofline readings, the random-line fraction, parity against the oracle."""
    import copy

    a = copy.copy(args)
    a.lists = "haplotypes"
    a.min_timed_s = args.realistic_timed_s
    a.no_streaming = True
    a.calibrate = False
    a.cpu_seconds = min(args.cpu_seconds, 2.0)
    a.parity_reads = min(args.parity_reads, 1024)
    sub = run_classify(a, np, kmers, lib, check, _lib, dev, dist, 1, 0, placement)
    # the same treatment as the headline's `traffic`: a counter pass of this very workload, now, as a child process
    note = live_traffic(a, sub)
    r = sub["roofline"]
    r.pop("_pricing", None)
    if note:
        r["traffic_source"] = f"{r.get('traffic_source')} [{note}]"
    keep = {
        "lists": "haplotypes: " + sub["data"], "value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"],
        "kernel_resident": (sub.get("kernel_resident") or {}).get("gbases_per_s"),
        "kernel_ms_avg": r["kernel_ms_avg"], "whole_probe_ms_avg": r["whole_probe_ms_avg"],
        "frac": r["frac"], "frac_P2_two_probe_reading": r["frac_P2_two_probe_reading"], "achieved": r["achieved"],
        "traffic": r["traffic"], "traffic_source": r["traffic_source"], "random_line_frac": r.get("random_line_frac"),
        "random_lines_Gps": r.get("random_lines_Gps"), "random_line_frac_same_table": r.get("random_line_frac_same_table"),
        "random_line_ceiling_same_table_Gps": (r.get("random_line_ceiling_same_table_Gps") or {}).get("best"),
        "kmers_per_list": sub["config"]["kmers_per_list"], "table_bytes_per_gpu": sub["config"]["table_bytes_per_gpu"],
        "table_bytes_per_key": sub["config"]["table_bytes_per_key"], "line_layout": sub["config"]["line_layout"],
        "bucket_select": sub["config"]["bucket_select"], "keys_behind_front": sub["config"]["keys_behind_front"],
        "layout_builds": sub["config"]["layout_builds"],
        "timed_regions": sub["timed_regions"], "timed_total_s": sub["timed_total_s"], "parity": sub["parity"], "bins": sub["bins"],
        "cpu_baseline_1_thread_gbases_per_s": (sub.get("cpu_baseline") or {}).get("value"),
        "note": "same steps, same kernels, same host-fed stage as `value`; only the lists (and the reads drawn from the two haplotypes) differ",
    }
    return keep


def kernel_fingerprint():
    """sha256 over the machine code of the probe kernels: the `.text` section of the gfx950 code object inside
    trio_binning_amd/csrc/build/tbk_kernels.o (read with a few lines of ELF / offload-bundle parsing: no tool, no child
    process).  A PMC traffic record (profiles/pmc_traffic*.json) is only replayed into `roofline.traffic` when it was
    taken on these very kernels; a comment in the source changes nothing, an instruction does.  (The fat binary as a
    whole is not reproducible from one compile to the next - the device code is.)  Falls back to the sources' hash
    when the object is not there."""
    import hashlib
    import struct

    def sections(buf, base):
        if buf[base:base + 4] != b"\x7fELF":
            raise ValueError("not an ELF object")
        shoff, = struct.unpack_from("<Q", buf, base + 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", buf, base + 0x3A)
        hdr = lambda i: struct.unpack_from("<IIQQQQIIQQ", buf, base + shoff + i * shentsize)
        stro = base + hdr(shstrndx)[4]
        out = {}
        for i in range(shnum):
            h = hdr(i)
            name = buf[stro + h[0]:buf.index(b"\0", stro + h[0])].decode()
            out[name] = (base + h[4], h[5])
        return out

    try:
        buf = open(os.path.join(ROOT, "trio_binning_amd", "csrc", "build", "tbk_kernels.o"), "rb").read()
        off, size = sections(buf, 0)[".hip_fatbin"]
        fb = buf[off:off + size]
        if fb[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
            raise ValueError("unexpected offload bundle")
        n, = struct.unpack_from("<Q", fb, 24)
        p = 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", fb, p)
            triple = fb[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and sz:
                to, ts = sections(fb, o)[".text"]
                return "text:" + hashlib.sha256(fb[to:to + ts]).hexdigest()
        raise ValueError("no gfx950 code object")
    except Exception:
        h = hashlib.sha256()
        for name in ("tbk_kernels.hip", "tbk_common.h", "tbk_device.h"):
            with open(os.path.join(ROOT, "trio_binning_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
        return "source:" + h.hexdigest()


def kernel_resources(stats):
    """Registers, LDS and the waves per SIMD they allow, of the single-read probe kernel this table runs, read from the
    gfx950 code object in build/tbk_kernels.o (the AMDGPU metadata note: what the loader allocates - rocprofv3's VGPR column
    is not the allocation).  None when the object or the kernel is not there."""
    import struct

    try:
        import msgpack

        buf = open(os.path.join(ROOT, "trio_binning_amd", "csrc", "build", "tbk_kernels.o"), "rb").read()

        def sections(b, base):
            shoff, = struct.unpack_from("<Q", b, base + 0x28)
            shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, base + 0x3A)
            hdr = lambda i: struct.unpack_from("<IIQQQQIIQQ", b, base + shoff + i * shentsize)
            stro = base + hdr(shstrndx)[4]
            return {b[stro + hdr(i)[0]:b.index(b"\0", stro + hdr(i)[0])].decode(): (base + hdr(i)[4], hdr(i)[5]) for i in range(shnum)}

        off, size = sections(buf, 0)[".hip_fatbin"]
        fb = buf[off:off + size]
        n, = struct.unpack_from("<Q", fb, 24)
        p, meta = 32, None
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", fb, p)
            triple = fb[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and sz:
                no, ns = sections(fb, o)[".note"]
                q = no
                while q + 12 <= no + ns:
                    namesz, descsz, ntype = struct.unpack_from("<III", fb, q)
                    q += 12
                    name = fb[q:q + namesz].rstrip(b"\0")
                    q += (namesz + 3) // 4 * 4
                    if name == b"AMDGPU" and ntype == 32:
                        meta = msgpack.unpackb(fb[q:q + descsz], raw=False, strict_map_key=False)
                    q += (descsz + 3) // 4 * 4
        w, m, t = stats["minimizer_w"], stats["minimizer_m"], stats["sampling_t"]
        if stats.get("entry_layout") or stats.get("short_keys") or stats.get("full_keys"):
            kind = 2 if stats.get("short_keys") else 3 if stats.get("full_keys") else 1 if stats.get("wide_entries") else 0
            lw = 3 if t and t == m - 2 * w else 2
            want = f"_Z22tbk_probe_entry_kernelILi{w}ELb0ELb0ELi{kind}ELi{lw}EEv9ProbeArgs"
        else:
            b = lambda x: "Lb1" if x else "Lb0"
            want = f"_Z16tbk_probe_kernelILi{w}E{b(m > 16)}E{b(bool(t))}E{b(stats.get('front_layout'))}ELb0ELb0EEv9ProbeArgs"
        for kern in meta["amdhsa.kernels"]:
            if kern[".name"] == want:
                vgpr = kern[".vgpr_count"] + kern.get(".agpr_count", 0)
                lds = kern[".group_segment_fixed_size"]
                by_regs = min(8, 512 // max(8, (vgpr + 7) // 8 * 8))
                by_lds = (160 * 1024 // max(1, lds)) // 4 if lds else 8  # one-wave blocks: a CU's 160 KB of LDS over its four SIMDs
                return {"kernel_symbol": want, "vgpr_count": vgpr, "sgpr_count": kern.get(".sgpr_count"), "lds_bytes_per_block": lds, "scratch_bytes_per_lane": kern.get(".private_segment_fixed_size"),
                        "vgpr_spill_count": kern.get(".vgpr_spill_count"), "waves_per_simd": min(by_regs, by_lds, 8),
                        "source": "AMDGPU metadata note of the gfx950 code object in trio_binning_amd/csrc/build/tbk_kernels.o"}
        return None
    except Exception:
        return None


def is_single_read_probe(name):
    """The dominant kernel by its demangled name: tbk_probe_kernel<W, M64, SAMP, FRONT, MULTI = false> or the entry kernels'
    tbk_probe_entry_kernel<W, MULTI = false, TWO = false, LW, KIND> (passes inside one read)."""
    import re

    n = name.replace(" ", "")
    return ("tbk_probe_kernel" in n and n.endswith("false>(ProbeArgs)")) or re.search(r"tbk_probe_entry_kernel<\d+,false,false,\d+(,\d+)?>\(ProbeArgs\)", n) is not None


def counter_means(csv_files):
    """Per-launch means of a rocprofv3 --pmc pass (its *_counter_collection.csv) over the FULL-SIZE launches of the single-read
    kernel: a bench run also classifies the parity reads (small launches), which must not dilute a per-launch figure.
    -> ({counter: mean}, {grid, launches, the row's register columns})"""
    import collections
    import csv

    rows = []
    for f in csv_files:
        rows += [r for r in csv.DictReader(open(f)) if is_single_read_probe(r["Kernel_Name"])]
    if not rows:
        return {}, {}
    full = max(int(r["Grid_Size"]) for r in rows)
    agg, meta = collections.defaultdict(list), {}
    for r in rows:
        if int(r["Grid_Size"]) == full:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"VGPR_Count_as_rocprofv3_reports_it": r.get("VGPR_Count"), "SGPR_Count": r.get("SGPR_Count"), "LDS_Block_Size": r.get("LDS_Block_Size"),
                    "Grid_Size": r.get("Grid_Size"), "Kernel_Name": r["Kernel_Name"][:80], "launches_averaged": len(agg[r["Counter_Name"]])}
    return {k: sum(v) / len(v) for k, v in agg.items()}, meta


def live_traffic(args, out, extra_child_args=()):
    """`roofline.traffic` from counters of this run: the same workload for four steps, as a child process under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (counters in a pass of their own, the program right behind `--`, cwd /tmp:
    MI355X_MICROARCH.md's HBM recipe), started when the timed legs are over and their device memory is released.  HBM bytes
    of one full-size launch of the single-read kernel = FETCH_SIZE [KiB] x 1024 x 2 (gfx950 tallies a 128-byte request at 64).
    Returns a note for `traffic_source` when the pass could not be made (the replayed record, if any, then stands)."""
    import glob
    import shutil
    import subprocess
    import tempfile

    if args.live_pmc == "off":
        return "live counter pass switched off (--live-pmc off)"
    if "--lists" not in extra_child_args and args.lists != "uniform":
        extra_child_args = ["--lists", args.lists] + list(extra_child_args)
    if args.live_pmc == "auto":
        if args.scaling != "weak" or args.read_lengths != "fixed" or args.rings > 1:
            return "no live counter pass for this workload (--live-pmc on asks for one)"
        if any(key.startswith(("ROCPROF", "ROCP_")) for key in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            return "no live counter pass: this run is itself under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.isfile("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return "no live counter pass: rocprofv3 not found"
    skip, child_args, it = {"--gpus": 1, "--steps": 1, "--warmup": 1, "--min-timed-s": 1, "--live-pmc": 1, "--cpu-seconds": 1,
                            "--no-cpu-baseline": 0, "--no-streaming": 0, "--no-realistic": 0, "--no-sweep": 0, "--no-strong-leg": 0, "--calibrate": 0, "--lists": 1}, [], iter(sys.argv[1:])
    for a in it:
        name = a.split("=", 1)[0]
        if name in skip:
            if skip[name] and "=" not in a:
                next(it, None)
            continue
        child_args.append(a)
    child_args += list(extra_child_args)   # (the realistic_lists sub-record's pass: --lists haplotypes)
    child_args += ["--steps", "4", "--warmup", "1", "--min-timed-s", "0", "--no-cpu-baseline", "--no-streaming", "--no-realistic", "--no-sweep", "--no-strong-leg",
                   "--live-pmc", "off"]   # (the timed path stays the parent's: host-fed batches are read packed, resident ones as ASCII)
    tmp = tempfile.mkdtemp(prefix="tbk_live_pmc_", dir="/tmp")
    t0 = time.time()
    try:
        env = dict(os.environ, TMPDIR="/tmp", TBK_SKIP_BUILD="1")
        cmd = [exe, "--kernel-trace", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.join(ROOT, "bench.py")] + child_args
        # (a process group of its own: on a timeout the profiler AND the bench under it are ended, by the group's id)
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            c_out, c_err = proc.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            import signal

            os.killpg(proc.pid, signal.SIGKILL)
            proc.communicate()
            return "live counter pass timed out after 150 s"
        run = argparse.Namespace(returncode=proc.returncode, stdout=c_out, stderr=c_err)
        line = next((ln for ln in reversed(run.stdout.splitlines()) if ln.startswith('{"metric"')), None)
        means, meta = counter_means(sorted(glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True)))
        if run.returncode != 0 or line is None or "FETCH_SIZE" not in means:
            return f"live counter pass failed (rocprofv3 exit {run.returncode}, counters {sorted(means)}): {run.stderr[-300:]!r}"
        child = json.loads(line)
        rf, crf = out["roofline"], child["roofline"]
        if crf["windows_per_launch"] != rf["windows_per_launch"] or child["config"]["table_bytes_per_gpu"] != out["config"]["table_bytes_per_gpu"]:
            return "live counter pass ran another launch shape than the timed legs; not used"
        traffic = means["FETCH_SIZE"] * 1024 * 2
        rf["traffic"] = int(traffic)
        rf["traffic_source"] = (f"measured in this run: when the timed legs were over, rank 0 ran this workload for four steps as a child process under `rocprofv3 --kernel-trace --pmc "
                                f"FETCH_SIZE` ({time.time() - t0:.0f} s; counters in a pass of their own) - FETCH_SIZE x 1024 x 2 (gfx950 tallies a 128-byte request at 64 bytes: "
                                f"MI355X_MICROARCH.md, HBM), mean over the {meta['launches_averaged']} full-size launches of the single-read kernel (grid {meta['Grid_Size']})")
        rf["traffic_fetch_size_kib_raw"] = round(means["FETCH_SIZE"], 1)
        price_traffic(rf, traffic)
        return None
    except Exception as e:  # a profiler that cannot run here must not cost the bench line
        return f"live counter pass failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def price_traffic(roofline, traffic):
    """HBM bytes of one launch of the dominant kernel against what the memory system delivers (PMC-measured bytes over this run's
    kernel time; 128-byte lines per second against the gather ceiling of profiles/calibration.json)."""
    pr = roofline["_pricing"]
    single_s, alg_bytes = pr["single_s"], pr["alg_bytes"]
    stats = {"table_bytes": pr["table_bytes"]}
    if traffic is not None and single_s > 0:
        # where the kernel sits against what the memory system can actually deliver: PMC-measured
        # bytes per launch over this run's kernel time, and 128-byte lines per second against the
        # random-line ceiling measured by tools/calib_footprint.py (profiles/calibration.json)
        roofline["traffic_GBps"] = round(traffic / single_s / 1e9, 1)
        roofline["traffic_frac_of_peak"] = round(traffic / single_s / 1e9 / HBM_PEAK_GBPS, 4)
        roofline["traffic_over_algorithmic"] = round(traffic / alg_bytes, 3)
        cfile = os.path.join(ROOT, "profiles", "calibration.json")
        if os.path.isfile(cfile):
            try:
                cal = json.load(open(cfile))
                # the calibration row whose footprint is closest to this table's
                # (round 5: the gather in the probe kernels' own shape - tbk_calib_gather_pairs, tools/calib_ceilings.py - where it was
                # measured; the best shape of the footprint nearest to this table's)
                rows = cal.get("random_lines_tuned_Glines_per_s") or cal["random_lines_Glines_per_s"]
                key = min(rows, key=lambda name: abs(float(name[:-2]) - stats["table_bytes"] / 1e9))
                ceiling = rows[key].get("best", rows[key].get("line128"))
                roofline["random_line_ceiling_footprint"] = key
                # random lines = what is not the read stream itself (the launch's bases as they lie in HBM: 0.25 B per base packed, 1 as ASCII)
                random_lines = max(0.0, traffic - pr.get("stream_bytes", 0.0)) / 128
                roofline["read_stream_bytes_per_launch"] = int(pr.get("stream_bytes", 0))
                roofline["random_lines_Gps"] = round(random_lines / single_s / 1e9, 2)
                roofline["random_line_ceiling_Gps"] = ceiling
                roofline["random_line_frac"] = round(random_lines / single_s / 1e9 / ceiling, 3)
                roofline["measured_stream_GBps"] = cal.get("stream_tuned_GBps", cal["guide_stream_GBps"])
                roofline["traffic_frac_of_measured_stream"] = round(traffic / single_s / 1e9 / roofline["measured_stream_GBps"], 3)
                if pr.get("same_table"):
                    # ... and against the same gather over this run's own table, measured in this run behind the timed legs
                    roofline["random_line_ceiling_same_table_Gps"] = pr["same_table"]
                    roofline["random_line_frac_same_table"] = round(random_lines / single_s / 1e9 / pr["same_table"]["best"], 3)
                roofline["random_line_note"] = ("the ceiling is what a pure gather in the probe kernels' own request shape sustains at this footprint (one-wave blocks, two lanes x 16 bytes "
                                                "of a line, the best of 1-8 lines in flight per pair and 4-8 waves per SIMD: tools/calib_ceilings.py) on the box of the reference run; boxes of the pool differ "
                                                "by +-4 % and placements of a table on one box by +-2 %, so `random_line_frac_same_table` prices the kernel against that gather over "
                                                "this run's own table (2^28 random lines per shape, behind the timed legs)")
            except Exception:
                pass



def run_classify(args, np, kmers, lib, check, _lib, dev, dist, world, rank, placement):
    k, n_list, L, R = args.k, args.kmers_per_list, args.read_len, args.reads_per_step

    def dalloc(nbytes):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, nbytes, C.byref(p)))
        return p.value

    t_setup = time.time()
    # ---- tables: 2 x n_list distinct canonical k-mers, generated and inserted on the GPU ----
    hap = args.lists == "haplotypes"
    genome_len = snp24 = err24 = 0
    if hap:
        # genome long enough for ~n_list windows that cover a position where the haplotypes differ
        snp24 = min((1 << 24) - 1, max(1, int(round(args.snp_rate * (1 << 24))))) | (min(255, int(round(args.repeat_fraction * 256))) << 24)
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
    # the library's pipeline over this rank's device: it hashes both lists into the paired table in HBM and
    # starts the device's feeder thread; `cls` is its classifier (the resident leg and the timing read it
    # directly while the pipeline is idle)
    pipe = kmers.MultiClassifier(hap_a, hap_b, [dev] * max(1, args.rings))
    cls = pipe._part(0)
    check(lib.tbk_device_sync(dev))
    t_build = time.time() - t0
    stats = cls.stats()
    if not hap:
        assert stats["distinct_a"] == n_list and stats["distinct_b"] == n_list, stats

    sweep_rec = None
    if args.sweep and rank == 0:
        from trio_binning_amd.sweep import full_membership_sweep

        t_sw = time.time()
        sweep_rec = full_membership_sweep(cls, hap_a, hap_b, hap_a.device_keys, hap_b.device_keys, n_list, n_list, k, device=dev,
                                          uniform_seed=None if hap else KEY_SEED)
        sweep_rec["seconds"] = round(time.time() - t_sw, 1)
        sweep_rec["what"] = ("every key of both lists as a read of k bases and packed 512 to a long read, as many non-members, every key with one base "
                             "substituted: counts compared on the device with what the lists say (c/kmers.c:112-122, 245-268)")

    want_cpu = rank == 0 and not args.no_cpu_baseline
    h_keys = None
    if want_cpu:
        h_keys = np.empty(2 * n_list, dtype=np.uint64)
        check(lib.tbk_memcpy_d2h(dev, h_keys.ctypes.data, C.c_void_p(d_keys), n_list * 8))
        check(lib.tbk_memcpy_d2h(dev, h_keys.ctypes.data + n_list * 8, C.c_void_p(d_keys + key_stride * 8), n_list * 8))
    check(lib.tbk_device_free(dev, C.c_void_p(d_keys)))

    ragged = args.read_lengths == "lognormal"

    def lengths_of(first, n_r):
        """offsets[0 .. n_r] of reads first .. first + n_r - 1 of the generator (host array)"""
        if not ragged:
            return np.arange(n_r + 1, dtype=np.uint64) * np.uint64(L)
        offs = np.zeros(n_r + 1, dtype=np.uint64)
        check(lib.tbk_synth_lognormal_lengths(READ_SEED, first, n_r, args.n50, args.sigma, args.short_fraction, 1, args.max_read_len, offs.ctypes.data))
        return offs

    def synth_reads(first, n_r, d_bases, d_offs, offs=None):
        if ragged:
            offs = lengths_of(first, n_r) if offs is None else offs
            tot = int(offs[-1])
            check(lib.tbk_memcpy_h2d(dev, C.c_void_p(d_offs), offs.ctypes.data, offs.nbytes))
            if hap:
                check(lib.tbk_synth_hap_reads_ragged_device(dev, KEY_SEED, genome_len, snp24, READ_SEED, first, n_r, C.c_void_p(d_offs), tot,
                                                            int(np.diff(offs).max()), err24, C.c_void_p(d_bases)))
            else:
                check(lib.tbk_synth_reads_ragged_device(dev, READ_SEED, first, n_r, C.c_void_p(d_offs), tot, KEY_SEED, n_list, n_list, k,
                                                        max(k, 15_000 // max(1, args.plant_major + args.plant_minor)), C.c_void_p(d_bases)))
        elif hap:
            check(lib.tbk_synth_hap_reads_device(dev, KEY_SEED, genome_len, snp24, READ_SEED, first, n_r, L, err24,
                                                 C.c_void_p(d_bases), C.c_void_p(d_offs)))
        else:
            check(lib.tbk_synth_reads_device(dev, READ_SEED, first, n_r, L, KEY_SEED, n_list, n_list, k,
                                             args.plant_major, args.plant_minor, C.c_void_p(d_bases), C.c_void_p(d_offs)))

    # ---- reads ------------------------------------------------------------------------------------
    # weak: `resident_batches` batches per rank, cycled; read indices disjoint between ranks.
    # strong: the rank's shard [lo, hi) of ONE fixed set of reads in batches of at most R reads; the
    # generator is indexed by absolute read number, so the set does not depend on how many ranks share it.
    # Every batch exists twice: in pinned host memory in the packed transfer format (what the reader hands
    # over; the host-fed path's input) and, for the resident leg, as ASCII in HBM.
    strong = args.scaling == "strong"
    if strong:
        lo, hi = shard_plan(args.strong_reads, rank, world)
        spans = [(s, min(s + R, hi)) for s in range(lo, hi, R)]
    else:
        nb = max(1, args.resident_batches)
        spans = [((rank * nb + b) * R, (rank * nb + b + 1) * R) for b in range(nb)]
    if not spans:
        raise SystemExit(f"rank {rank}: empty shard (more ranks than reads?)")
    host_fed = args.timed_path == "host_fed"
    keep_resident = (not strong) or not host_fed  # a strong shard is kept in HBM only when that is what is timed
    r_cap = max(last - first for first, last in spans)
    span_offs = [lengths_of(first, last - first) for first, last in spans]
    b_cap = max(int(o[-1]) for o in span_offs)
    stage = kmers.pinned_empty((b_cap,), np.uint8)  # ASCII on its way from the generator (HBM) to the packer
    batches = []  # (d_bases, d_offsets, n_reads, total_bases, packed host batch)
    windows_of = []  # window starts of every batch: sum of max(0, length - k + 1)
    d_bases = d_offs = None
    for (first, last), offs in zip(spans, span_offs):
        n_r = last - first
        tot = int(offs[-1])
        if d_bases is None or keep_resident:
            d_bases = dalloc(((b_cap if strong else tot) + 15) // 16 * 16 + 16)
            d_offs = dalloc((r_cap + 1) * 8)
        synth_reads(first, n_r, d_bases, d_offs, offs)
        check(lib.tbk_memcpy_d2h(dev, stage.ctypes.data, C.c_void_p(d_bases), tot))
        packed = kmers.pack_bases(stage[:tot], offs, pinned=True)
        batches.append((d_bases if keep_resident else None, d_offs if keep_resident else None, n_r, tot, packed))
        windows_of.append(int(np.maximum(np.diff(offs).astype(np.int64) - k + 1, 0).sum()))
    del stage
    t_setup = time.time() - t_setup

    # batches the bench keeps submitted: the rings' slots plus one waiting per ring (what the pipeline admits), so
    # that a feeder whose ring has just got room finds its next batch queued already
    depth = pipe.depth + len(pipe.devices)
    counts_ring = [kmers.pinned_empty((r_cap, 2), np.int32) for _ in range(depth)]
    num_a, num_b = hap_a.num_kmers, hap_b.num_kmers
    bins_total = {"A": 0, "B": 0, "U": 0}

    def finish(waiter, ticket, slot, n_r, tally):
        """Host side of a step: wait for its counts, take the A/B/U decision."""
        waiter(ticket)
        _, _, bins = kmers.score_and_bin(counts_ring[slot][:n_r], num_a, num_b)
        if tally:
            for name, ch in (("A", b"A"), ("B", b"B"), ("U", b"U")):
                bins_total[name] += bins.count(ch)

    # a step: weak = one batch (cycled); strong = every batch of the shard
    launches_per_step = len(batches) if strong else 1
    bases_per_step = sum(b[3] for b in batches) if strong else batches[0][3]

    def run(n_steps, tally, fed):
        """n_steps steps, pipelined: the host finishes step i-1 while the device works on step i.
        fed: through the pipeline from pinned host memory; else: batches resident in HBM."""
        pending = []
        waiter = pipe.wait if fed else cls.wait
        # host-fed: as many batches submitted as the ring holds (the third one's copy runs beside the first one's
        # kernel); resident: one fewer is enough to keep the compute stream busy
        ahead = depth if fed else max(1, cls.depth - 1)  # (the resident leg drives ring 0's classifier alone, its own ring of 3)
        for i in range(n_steps * launches_per_step):
            d_b, d_o, n_r, tot, packed = batches[i % len(batches)]
            slot = i % depth
            if len(pending) == ahead:
                finish(waiter, *pending.pop(0), tally)
            if fed:
                t = pipe.submit_packed(packed, counts_ring[slot][:n_r])
            else:
                t = cls.submit_device(d_b, d_o, n_r, tot, counts_ring[slot][:n_r])
            pending.append((t, slot, n_r))
        while pending:
            finish(waiter, *pending.pop(0), tally)

    def region_of(fed, tally):
        def region():
            check(lib.tbk_device_sync(dev))
            dist.barrier()
            t0 = time.perf_counter()
            run(args.steps, tally, fed)
            check(lib.tbk_device_sync(dev))
            dist.barrier()
            return time.perf_counter() - t0
        return region

    def timed(fed, tally):
        run(args.warmup, False, fed)
        cls.kernel_timing(True)
        region_s = timed_regions(region_of(fed, tally), dist, args.min_timed_s)
        launches, probe_ms, single_ms = cls.kernel_timing_read2()
        cls.kernel_timing(False)
        return region_s, launches, probe_ms, single_ms

    bases_all = dist.reduce(args.steps * bases_per_step, "SUM")
    region_s, launches, probe_ms, single_ms = timed(host_fed, True)
    n_passes, multi_passes = cls.last_passes()  # of the last probe (weak scaling: every batch has this shape)
    elapsed = statistics.median(region_s)
    value = bases_all / elapsed / 1e9
    other = None
    if keep_resident and host_fed:
        o_region_s, o_l, o_probe_ms, o_single_ms = timed(False, False)
        o_el = statistics.median(o_region_s)
        other = {"gbases_per_s": round(bases_all / o_el / 1e9, 3), "ms_per_step": round(o_el / args.steps * 1e3, 3),
                 "probe_ms_avg": round(o_probe_ms / max(1, o_l), 4), "single_read_kernel_ms_avg": round(o_single_ms / max(1, o_l), 4),
                 "note": "the same steps with the batches already in HBM (no H2D): what the kernels alone sustain"}

    # the ceiling of the window loop's line rate on THIS table (where its pages lie moves the rate by +-2 %: EXPERIMENTS.md): a pure
    # gather in the loop's own request shape over the table the timed legs have just used - 2^28 random lines (5 ms) per shape
    same_table = None
    if rank == 0 and hasattr(lib, "tbk_classifier_calibrate_pairs"):
        try:
            shapes = {f"waves8_inflight{inf}": round(max(cls.calibrate_pairs(inf, 8) for _ in range(2)) / 1e9, 2) for inf in (2, 4, 8)}
            same_table = {"best": max(shapes.values()), **shapes}
        except Exception as e:
            print(f"bench: same-table gather calibration failed: {e}", file=sys.stderr)

    if hasattr(lib, "tbk_debug_counters"):  # debug build (-DTBK_COUNTERS): event counts of one step, to stderr
        buf = (C.c_ulonglong * 8)()
        lib.tbk_debug_counters(buf, 1)
        run(1, False, host_fed)
        lib.tbk_debug_counters(buf, 1)
        w_ = launches_per_step * batches[0][2] * max(1, L - k + 1)
        print("tbk-counters", json.dumps({"careful_jstep_frac": round(buf[1] / max(buf[0], 1), 4), "careful_substeps_per_jstep": round(buf[2] / max(buf[0], 1), 4),
                                         "walks_per_window": round(buf[3] / w_, 6), "lines_per_window": round(buf[4] / 4 / w_, 4),
                                         "back_half_looks_per_window": round(buf[5] / w_, 5)}), file=sys.stderr)

    if not stats["minimizer_w"]:
        bucket_select = "plain hash"
    elif stats["sampling_t"]:
        bucket_select = "mod-sampling w=%d m=%d t=%d" % (stats["minimizer_w"], stats["minimizer_m"], stats["sampling_t"])
    else:
        bucket_select = "minimizer w=%d m=%d" % (stats["minimizer_w"], stats["minimizer_m"])

    # ---- parity: every rank classifies the generator's first reads through the host-fed path ------------
    n_par = max(1, min(args.parity_reads, 1 << 16))
    par_offs = lengths_of(0, n_par)
    par_total = int(par_offs[-1])
    d_pb, d_po = dalloc((par_total + 15) // 16 * 16 + 16), dalloc((n_par + 1) * 8)
    synth_reads(0, n_par, d_pb, d_po, par_offs)
    par_bases = kmers.pinned_empty((par_total,), np.uint8)
    check(lib.tbk_memcpy_d2h(dev, par_bases.ctypes.data, C.c_void_p(d_pb), par_bases.nbytes))
    par_counts = pipe.wait(pipe.submit(par_bases, par_offs)).copy()                      # ASCII in, packed by the feeder
    par_counts2 = pipe.wait(pipe.submit_packed(kmers.pack_bases(par_bases, par_offs)))   # packed ahead, as the reader does
    import zlib

    mine = {"rank": rank, "device_index": dev, "device": _lib.device_identity(dev), "placement": placement, "sum_a": int(par_counts[:, 0].sum()), "sum_b": int(par_counts[:, 1].sum()),
            "crc32": zlib.crc32(par_counts.tobytes()), "transfers_agree": bool(np.array_equal(par_counts, par_counts2))}
    everyone = dist.gather_obj(mine)
    parity = {"reads_checked_per_rank": n_par, "bases_checked_per_rank": par_total, "count_checksum": [mine["sum_a"], mine["sum_b"], mine["crc32"]],
              "all_ranks_equal": all((e["sum_a"], e["sum_b"], e["crc32"]) == (mine["sum_a"], mine["sum_b"], mine["crc32"]) for e in everyone),
              "packed_and_ascii_transfers_agree": all(e["transfers_agree"] for e in everyone)}
    devices = [{"rank": e["rank"], "device_index": e["device_index"], "id": e["device"], **e["placement"]} for e in everyone]

    # ---- roofline of the dominant kernel (rank 0's device) ----------------------------------------------
    # The probe is four kernels: pass index, the multi-read and two-read kernels (passes that touch several reads) and the
    # single-read kernel (passes inside one read: on 15 kb reads 86 % of the passes).  The roofline is the
    # single-read kernel's: its windows x the algorithmic bytes per window (1 read byte + 8 B for
    # the hapA slot + 8 B for the hapB slot when hapA missed; P = 1 reading: 1 + 8) / its HIP-event time.
    hit_a_frac = float(par_counts[:, 0].sum()) / max(1, int(np.maximum(np.diff(par_offs).astype(np.int64) - k + 1, 0).sum()))
    reads_per_launch = sum(b[2] for b in batches) / len(batches)
    bases_per_launch = sum(b[3] for b in batches) / len(batches)
    windows = sum(windows_of) / len(batches)
    single_frac = 1.0 - multi_passes / max(1, n_passes)
    windows_single = windows * single_frac
    # "a merged single-table implementation must report with P = 1 (B_alg = 9 B)".  The design is one paired
    # table probed once per window, so `frac` is the P = 1 reading; the two-probe reading
    # (1 + 8 + 8 per window that misses hapA) is printed beside it.
    b_alg = 9.0
    b_alg_p2 = 9 + 8 * (1 - hit_a_frac)
    alg_bytes = windows_single * b_alg
    single_s = single_ms / max(1, launches) * 1e-3
    probe_s = probe_ms / max(1, launches) * 1e-3
    achieved = alg_bytes / single_s / 1e9 if single_s > 0 else 0.0
    table_load = n_list / (stats["n_buckets"] * 8)
    traffic = traffic_src = None
    for tname in ("pmc_traffic.json", "pmc_traffic_haplotypes.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if not os.path.isfile(tfile):
            continue
        try:
            t = json.load(open(tfile))
            same = (t.get("read_len") == L and t.get("kmers_per_list") == n_list
                    and t.get("k") == k and t.get("bucket_select") == bucket_select and t.get("lists", "uniform") == args.lists
                    and abs(t.get("table_load", 0) - table_load) < 2e-3
                    and bool(t.get("front_layout", False)) == bool(stats.get("front_layout"))
                    and bool(t.get("entry_layout", False)) == bool(stats.get("entry_layout"))
                    and bool(t.get("short_keys", False)) == bool(stats.get("short_keys"))
                    and bool(t.get("full_keys", False)) == bool(stats.get("full_keys"))
                    and t.get("kernel") == "tbk_probe_kernel<single-read>")
            if same and t.get("kernel_sha256") != kernel_fingerprint():
                # taken on other kernels than the ones in this tree: stale bytes are not reported
                traffic_src = ("profiles/" + tname + " is STALE: its PMC passes ran on kernels " + str(t.get("kernel_sha256"))[:17]
                               + ", this tree's are " + kernel_fingerprint()[:17] + " - re-run tools/gpu_profile.sh; traffic withheld")
            elif same:  # measured in separate rocprofv3 --pmc passes on this configuration (not in this run); scaled to this launch's windows
                traffic = t["hbm_bytes_per_window"] * windows_single
                traffic_src = ("profiles/" + tname + " (rocprofv3 --pmc passes of this configuration on these very kernels, machine code sha256 "
                               + kernel_fingerprint()[5:17] + ", replayed per window; not measured in this run)")
        except Exception:
            pass
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None if traffic is None else int(traffic), "traffic_source": traffic_src,
        "kernel": ("tbk_probe_entry_kernel<W, MULTI=false, TWO=false>" if stats.get("entry_layout") else "tbk_probe_entry_kernel<W, MULTI=false, TWO=false, KIND=2: short keys>" if stats.get("short_keys")
                   else "tbk_probe_entry_kernel<W, MULTI=false, TWO=false, KIND=3: full keys>" if stats.get("full_keys")
                   else "tbk_probe_kernel<..., MULTI=false, TWO=false>") + " (single-read passes)", "kernel_ms_avg": round(single_s * 1e3, 4), "launches": int(launches),
        "timed_in": "the timed region of `value` (HIP events on the compute stream)",
        "alg_bytes_per_launch": int(alg_bytes), "alg_bytes_per_window": round(b_alg, 3), "windows_per_launch": int(windows_single),
        "share_of_the_batch_windows": round(single_frac, 4), "passes": int(n_passes), "multi_read_passes": int(multi_passes),
        "reading": "P = 1 (SURVEY 8d's merged-table rule: 1 read byte + one 8-byte slot per window)",
        "frac_P2_two_probe_reading": round(windows_single * b_alg_p2 / single_s / 1e9 / HBM_PEAK_GBPS, 4) if single_s > 0 else None,
        "alg_bytes_per_window_P2": round(b_alg_p2, 3),
        "whole_probe_ms_avg": round(probe_s * 1e3, 4),
        "whole_probe": {"what": "pass index + multi-read, two-read and single-read kernels, per batch", "alg_bytes": int(windows * b_alg),
                        "achieved": round(windows * b_alg / probe_s / 1e9, 1) if probe_s > 0 else None,
                        "frac": round(windows * b_alg / probe_s / 1e9 / HBM_PEAK_GBPS, 4) if probe_s > 0 else None},
        "kernel_only_gbases_per_s": round(bases_per_launch / probe_s / 1e9, 2) if probe_s > 0 else None,
    }
    roofline["kernel_resources"] = kernel_resources(stats)
    roofline["_pricing"] = {"single_s": single_s, "alg_bytes": alg_bytes, "table_bytes": stats["table_bytes"], "same_table": same_table,
                            "stream_bytes": bases_per_launch * single_frac * (sum(b[4].nbytes for b in batches) / max(1, sum(b[3] for b in batches)) if host_fed else 1.0)}   # (price_traffic's inputs; main() drops it)
    price_traffic(roofline, traffic)

    if want_cpu and L > 2_000_000:
        # the CPU sample works in whole reads; one read of this length is minutes of oracle time
        print(f"bench: reads of {L} bases are too long for a bounded CPU sample: cpu_baseline skipped", file=sys.stderr)
        want_cpu = False
    fed_txt = ("batches in pinned host memory in the packed transfer format (what the reader hands over), through the pipeline: H2D on the side stream, "
               "kernels, D2H, host binning" if host_fed else "batches resident in HBM")
    if ragged:
        lens_all = np.concatenate([np.diff(o) for o in span_offs]).astype(np.int64)
        order = np.sort(lens_all)[::-1]
        n50_got = int(order[np.searchsorted(np.cumsum(order), lens_all.sum() / 2)])
        read_lengths = {"distribution": f"log-normal, N50 {args.n50:g}, sigma {args.sigma:g}, {args.short_fraction:g} of the reads debris of 1 b .. 5 kb, clamped to {args.max_read_len}",
                        "reads": int(lens_all.size), "bases": int(lens_all.sum()), "n50": n50_got, "mean": round(float(lens_all.mean()), 1), "median": int(np.median(lens_all)),
                        "min": int(lens_all.min()), "max": int(lens_all.max()), "reads_over_1Mb": int((lens_all > 1_000_000).sum()), "reads_under_k": int((lens_all < k).sum())}
        workload = (f"BASELINE configs[4] shape on one GPU per rank: synthetic reads with log-normal lengths (N50 {n50_got}, {lens_all.size} reads = {lens_all.sum() / 1e9:.2f} Gbp "
                    f"{'in all, split over ' + str(world) + ' rank(s) by read index' if strong else 'per rank, cycled'}), 2x{n_list} unique {k}-mers replicated per GPU; {fed_txt}")
    elif strong:
        workload = (f"BASELINE configs[3]: one fixed set of {args.strong_reads} synthetic {L} b reads ({args.strong_reads * L / 1e9:.1f} Gbp) "
                    f"split over {world} rank(s) by read index, 2x{n_list} unique {k}-mers replicated per GPU; {fed_txt}; "
                    f"a step = one pass of every rank over its shard")
    else:
        workload = (f"BASELINE configs[2] shape: {L} b synthetic reads, 2x{n_list} unique {k}-mers replicated per GPU, "
                    f"{R} reads ({R * L / 1e9:.3f} Gbases) per step per GPU, reads sharded over ranks; {fed_txt}")
    out = {
        "metric": metric_label(k, n_list), "value": round(value, 3), "unit": "Gbases/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u64",
        "data": "synthetic" if not hap else f"synthetic haplotypes (SNP rate {args.snp_rate:g}, read error rate {args.error_rate:g}" + (f", {args.repeat_fraction:g} of the genome in repeats" if args.repeat_fraction else "") + ")",
        "config": {
            "workload": workload, "timed_path": args.timed_path,
            "k": k, "kmers_per_list": n_list, "read_len": L if not ragged else None, "read_lengths": read_lengths if ragged else "fixed", "reads_per_step": R if not strong else None,
            "distinct_batches": len(batches), "launches_per_step": launches_per_step, "bases_per_step_per_rank": bases_per_step,
            "h2d_bytes_per_base": round(sum(b[4].nbytes for b in batches) / max(1, sum(b[3] for b in batches)), 4) if host_fed else 0,
            "table_bytes_per_gpu": stats["table_bytes"], "table_bytes_per_key": round(stats["table_bytes"] / max(1, 2 * n_list), 1), "table_load": round(table_load, 4),
            "bucket_select": bucket_select, "lists": args.lists,
            "layout_builds": stats.get("layout_builds"), "keys_past_their_half": stats.get("keys_past_half"),
            "line_layout": (("entries (wide, 16 bytes)" if stats.get("wide_entries") else "entries") + ": a run of overlapping list k-mers stored once; 32 of a line's 128 bytes asked for per window, two lanes" if stats.get("entry_layout")
                            else "short keys: a list k-mer in 32 bits (what its bucket does not say already); 32 of a line's 128 bytes asked for per window (seven keys and the line's summary), two lanes" if stats.get("short_keys")
                            else "full keys: 64-bit keys, sixteen slots to a line; 32 of a line's 128 bytes asked for per window (three keys and the line's summary), two lanes" if stats.get("full_keys")
                            else "front: 64 of a line's 128 bytes asked for per window" if stats.get("front_layout") else "whole lines"),
            "keys_behind_front": stats.get("keys_behind_front"),
            "entries": [stats.get("entries_a"), stats.get("entries_b")] if stats.get("entry_layout") else None,
            "parallelism": f"read-sharded x{world}, tables replicated, no data-path collective",
            "rings_per_rank": max(1, args.rings), "batches_per_ring": pipe.dealt,
        },
        "timed_regions": len(region_s), "timed_total_s": round(sum(region_s), 3),
        "region_s_min_median_max": [round(min(region_s), 4), round(elapsed, 4), round(max(region_s), 4)],
        ("kernel_resident" if host_fed else "host_fed"): other,
        "roofline": roofline, "parity": parity, "devices": devices, "distinct_devices": len({d["id"] for d in devices}),
        "bins": bins_total, "setup_s": round(t_setup, 2), "table_build_s": round(t_build, 2),
        "device": _lib.device_name(dev),
    }
    if args.share_device:
        out["devices_note"] = "--share-device: every rank uses device 0 (a plumbing run, not a scaling result)"

    # ---- the same stage fed differently (rank 0, N = 1) ----------------------------------------------
    if rank == 0 and world == 1 and not args.no_streaming and not ragged:
        out["pipeline_variants"] = pipeline_variants(args, np, kmers, lib, check, dev, pipe, par_bases, par_offs, par_counts, batches[0], L)
    if rank == 0 and world == 1 and not args.no_streaming and not ragged and not hap and hasattr(lib, "tbk_gzip_bench_device"):
        try:
            out["bins_gzip_encoder"] = gzip_encoder_leg(np, kmers, lib, check, dev, par_bases, par_offs)
        except Exception as e:  # (a side record: it must not cost the bench line)
            print(f"bench: bins_gzip_encoder leg failed: {e}", file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_streaming and not ragged and not hap and hasattr(lib, "tbk_bgzf_bench_device"):
        try:
            out["input_bgzf_inflater"] = bgzf_inflater_leg(np, lib, check, dev, par_bases, par_offs)
        except Exception as e:
            print(f"bench: input_bgzf_inflater leg failed: {e}", file=sys.stderr)

    if args.calibrate and rank == 0:
        out["calibration"] = calibrate(lib, check, dev, stats["table_bytes"])

    # ---- CPU baseline + read-for-read parity on a bounded sample (rank 0, every N) ---------------------
    if want_cpu:
        out["cpu_baseline"], cpu_par = cpu_baseline(args, np, lib, par_bases, par_offs, par_counts, h_keys, n_list, k, L, n_par)
        out["parity"].update(cpu_par)
    if sweep_rec is not None:
        out["sweep"] = sweep_rec

    pipe.close()
    hap_a.close()
    hap_b.close()
    return out


def gzip_encoder_leg(np, kmers, lib, check, dev, par_bases, par_offs):
    """What follows the classify stage in the reference's default mode: the three bins written through gzip (seq.py:132-134,
    classify_by_kmers.py:86-92).  The bin writer codes its members on the device (csrc/tbk_gdeflate.hip); this leg times that encoder
    by itself on the parity reads as FASTQ text with HiFi-like qualities, cut into members of 1 MiB as the writer cuts them:
    the three-deep ring from pinned memory (GB/s of text, against the 55 GB/s the link brings) and one job's kernels between HIP
    events.  Every member's round trip through zlib is tests/test_gpu_deflate.py's business, not this leg's."""
    rng = np.random.default_rng(0x5EED0006)
    n_reads = min(len(par_offs) - 1, 4096)
    lens = np.diff(par_offs[: n_reads + 1]).astype(np.int64)
    total = int(lens.sum())
    qual = np.clip(rng.normal(60, 15, total), 2, 93).astype(np.uint8)
    qual[rng.random(total) < 0.6] = 93
    qual += 33
    pieces = []
    for r in range(n_reads):
        lo, hi = int(par_offs[r]), int(par_offs[r + 1])
        pieces += [b"@read%09d c\n" % r, par_bases[lo:hi].tobytes(), b"\n+\n", qual[lo:hi].tobytes(), b"\n"]
    text = b"".join(pieces)
    member = 1 << 20
    n_members = (len(text) + member - 1) // member
    pinned = kmers.pinned_empty((len(text),), np.uint8)
    pinned[:] = np.frombuffer(text, dtype=np.uint8)
    mlens = (C.c_uint64 * n_members)(*[min(member, len(text) - i * member) for i in range(n_members)])
    ps, ks, ob = C.c_double(), C.c_double(), C.c_uint64()
    check(lib.tbk_gzip_bench_device(dev, C.c_void_p(pinned.ctypes.data), mlens, n_members, 24, C.byref(ps), C.byref(ks), C.byref(ob)))
    link = 55.0
    return {
        "what": "the bins' gzip members coded on the device (the reference's default output, seq.py:132-134): the parity reads as FASTQ text with HiFi-like "
                f"qualities, {n_members} members of 1 MiB per job, 24 jobs through the three-deep ring from pinned host memory",
        "text_MB_per_job": round(len(text) / 1e6, 1), "members_MB_per_job": round(ob.value / 1e6, 1), "ratio": round(ob.value / len(text), 4),
        "text_GB_per_s": round(len(text) / ps.value / 1e9, 2), "gbases_per_s_equivalent": round(total / ps.value / 1e9, 2),
        "kernels_only_text_GB_per_s": round(len(text) / ks.value / 1e9, 2), "kernels_ms_per_job": round(ks.value * 1e3, 3),
        "roofline": {"bound": "pcie", "achieved": round(len(text) / ps.value / 1e9, 2), "peak": link, "unit": "GB/s of text over the link", "frac": round(len(text) / ps.value / 1e9 / link, 3),
                     "hbm_frac_of_the_kernels": round((len(text) + ob.value) / ks.value / 1e9 / HBM_PEAK_GBPS, 4)},
    }


def bgzf_inflater_leg(np, lib, check, dev, par_bases, par_offs):
    """What precedes the classify stage when the reads come as .fastq.gz written by bgzip / htslib (the reference reads any .gz through
    gzip.open, seq.py:86-92): the reader inflates the file's blocks on the device (csrc/tbk_gdeflate.hip, second half).  This leg times
    that inflater by itself on one window: the parity reads as FASTQ text with HiFi-like qualities, cut into 60 000-byte blocks and
    deflated by zlib at level 6 (what bgzip writes) - the window's kernels between HIP events (input resident; every block's CRC-32
    checked on the device in the timed region) and the ring as the reader drives it, staging copy and both link crossings included."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    rng = np.random.default_rng(0x5EED0007)
    n_par = len(par_offs) - 1
    n_reads = 2 * min(n_par, 4096)   # (the parity reads twice, with qualities of their own: about the 3 800 blocks of one of the reader's windows)
    pieces, total = [], 0
    for r in range(n_reads):
        lo, hi = int(par_offs[r % n_par]), int(par_offs[r % n_par + 1])
        q = np.clip(rng.normal(60, 15, hi - lo), 2, 93).astype(np.uint8)
        q[rng.random(hi - lo) < 0.6] = 93
        pieces += [b"@read%09d c\n" % r, par_bases[lo:hi].tobytes(), b"\n+\n", (q + 33).tobytes(), b"\n"]
        total += hi - lo
    text = b"".join(pieces)

    def block(i):
        blk = text[i:i + 60000]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(blk) + c.flush()
        return struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk))

    with ThreadPoolExecutor(os.cpu_count() or 4) as pool:   # (zlib releases the GIL)
        data = b"".join(pool.map(block, range(0, len(text), 60000)))
    n_blocks = (len(text) + 59999) // 60000
    rs, ks, tb = C.c_double(), C.c_double(), C.c_uint64()
    check(lib.tbk_bgzf_bench_device(dev, data, len(data), 12, C.byref(rs), C.byref(ks), C.byref(tb)))
    assert tb.value == len(text)
    return {
        "what": "bgzf input inflated on the device (what the reader does with a .fastq.gz of bgzf blocks before the classify stage): the parity reads as FASTQ text with "
                f"HiFi-like qualities, {n_blocks} blocks of 60 000 bytes deflated by zlib at level 6, one wave per block, every block's CRC-32 checked on the device",
        "text_MB_per_window": round(len(text) / 1e6, 1), "deflated_MB_per_window": round(len(data) / 1e6, 1), "ratio": round(len(data) / len(text), 4), "blocks": n_blocks,
        "kernels_only_text_GB_per_s": round(len(text) / ks.value / 1e9, 2), "kernels_ms_per_window": round(ks.value * 1e3, 3),
        "ring_text_GB_per_s": round(len(text) / rs.value / 1e9, 2), "gbases_per_s_equivalent": round(total / rs.value / 1e9, 2),
        "bound": "the kernel's scalar instructions: 56 per symbol against a SIMD's one scalar issue per four cycles (profiles/r06/ginflate_gate_pmc.json); its bytes "
                 f"(read {round(len(data) / len(text), 2)} + write 1 per byte of text) would allow {round(HBM_PEAK_GBPS / (1 + len(data) / len(text)) / 1e3, 1)} TB/s",
        "hbm_frac_of_the_kernels": round((len(text) + len(data)) / ks.value / 1e9 / HBM_PEAK_GBPS, 4),
    }


def pipeline_variants(args, np, kmers, lib, check, dev, pipe, par_bases, par_offs, par_counts, batch0, L):
    """The same classify stage fed with ASCII batches instead of the reader's packed ones (rank 0, N = 1):
      ascii_in_packed_by_feeder   ASCII batches in pinned host memory; the device's feeder thread packs each
                                  (all host threads) before the copy: what a caller without the reader gets;
      ascii_over_pcie             ASCII crosses the link, the kernel packs (TBK_PACKED_H2D=0): PCIe-bound.
    Three host batches are cycled through the pipeline for ~stream_seconds per variant; the counts must equal
    those of the packed path for the same reads."""
    d_bases = batch0[0]
    n_r0 = batch0[2]
    if d_bases is None:
        return None
    Rs = min(args.stream_batch_reads, n_r0 // 3 if n_r0 >= 3 else n_r0)
    if Rs < 1:
        return None
    tot = Rs * L
    host = []
    for b in range(min(3, n_r0 // Rs)):
        hb = kmers.pinned_empty((tot,), np.uint8)
        ho = kmers.pinned_empty((Rs + 1,), np.uint64)
        check(lib.tbk_memcpy_d2h(dev, hb.ctypes.data, C.c_void_p(d_bases + b * tot), tot))
        ho[:] = np.arange(Rs + 1, dtype=np.uint64) * np.uint64(L)
        want = pipe.wait(pipe.submit_packed(kmers.pack_bases(hb, ho))).copy()
        host.append((hb, ho, want))
    res = {"batch_reads": Rs, "batch_gbases": round(tot / 1e9, 4), "pinned_host_batches": len(host)}
    depth = pipe.depth

    def cycle(n_batches, check_counts):
        pend, same = [], True
        for i in range(n_batches):
            if len(pend) == depth:
                j, t = pend.pop(0)
                c = pipe.wait(t)
                same = same and (not check_counts or bool(np.array_equal(c, host[j][2])))
            j = i % len(host)
            pend.append((j, pipe.submit(host[j][0], host[j][1])))
        while pend:
            j, t = pend.pop(0)
            c = pipe.wait(t)
            same = same and (not check_counts or bool(np.array_equal(c, host[j][2])))
        return same

    def leg():
        cycle(3, False)
        t = time.perf_counter()
        cycle(6, False)
        per = (time.perf_counter() - t) / 6
        n = int(max(6, min(2000, args.stream_seconds / max(per, 1e-6))))
        t = time.perf_counter()
        same = cycle(n, True)
        dt = time.perf_counter() - t
        return n, dt, same

    part = pipe._part(0)
    part.packed_transfer = True
    n, dt, same = leg()
    res["ascii_in_packed_by_feeder"] = {"gbases_per_s": round(n * tot / dt / 1e9, 2), "h2d_GBps": round(n * (tot / 4 + (Rs + 1) * 8) / dt / 1e9, 2),
                                        "bytes_per_base": 0.25, "batches": n, "seconds": round(dt, 3), "counts_equal_packed_path": same,
                                        "host_pack_threads": int(lib.tbk_host_threads())}
    part.packed_transfer = False
    n, dt, same = leg()
    res["ascii_over_pcie"] = {"gbases_per_s": round(n * tot / dt / 1e9, 2), "h2d_GBps": round(n * (tot + (Rs + 1) * 8) / dt / 1e9, 2),
                              "bytes_per_base": 1.0, "batches": n, "seconds": round(dt, 3), "counts_equal_packed_path": same}
    part.packed_transfer = True
    return res


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


def cpu_baseline(args, np, lib, h_bases, offs, gpu_counts, h_keys, n_list, k, L, sample_reads):
    """The oracle on this box's host cores, same tables, the parity reads (the generator's first reads, which
    the GPU classified through the host-fed path)."""
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
    h_bases = np.asarray(h_bases)
    gpu_counts_batch0 = gpu_counts

    def timed(n_reads, threads):
        t = time.perf_counter()
        c = orc.count_batch(h_bases[: int(offs[n_reads])], offs[: n_reads + 1], oa, ob, threads=threads)
        return time.perf_counter() - t, c

    # 1 thread = what the reference does.  Calibrate on a few reads, then ~cpu_seconds worth.
    probe_n = min(sample_reads, 16)
    dt, _ = timed(probe_n, 1)
    n1 = int(max(probe_n, min(sample_reads, args.cpu_seconds / max(dt / probe_n, 1e-9))))
    dt1, c1 = timed(n1, 1)
    rate1 = int(offs[n1]) / dt1 / 1e9
    # all host cores, reads sharded over threads sharing the read-only tables
    nall = int(max(n1, min(sample_reads, n1 * cores * 0.7)))
    dtn, cn = timed(nall, cores)
    raten = int(offs[nall]) / dtn / 1e9
    # fairness datum (SURVEY 8d CPU-opt): rolling k-mers + all threads on the same tables; not the
    # reference's algorithm
    t = time.perf_counter()
    cf = orc.count_batch_fast(h_bases[: int(offs[sample_reads])], offs[: sample_reads + 1], oa, ob, threads=cores)
    dtf = time.perf_counter() - t
    ratef = int(offs[sample_reads]) / dtf / 1e9
    # parity: GPU counts of batch 0 (from the roofline step) vs the oracle on the sample
    g = gpu_counts_batch0[:nall]
    equal = (bool(np.array_equal(g, cn)) and bool(np.array_equal(g[:n1], c1))
             and bool(np.array_equal(gpu_counts_batch0[:sample_reads], cf)))
    parity = {"reads_checked_against_the_oracle": int(nall), "bases_checked_against_the_oracle": int(offs[nall]), "gpu_equals_cpu": equal,
              "cpu_count_sums": [int(cn[:, 0].sum()), int(cn[:, 1].sum())]}
    if not equal:
        bad = np.nonzero((g != cn).any(axis=1))[0][:5]
        parity["first_mismatches"] = [[int(i), g[i].tolist(), cn[i].tolist()] for i in bad]
    base = {
        "value": round(rate1, 6), "unit": "Gbases/s", "cores": 1, "kind": "port", "host_hardware_threads": hw_threads, "host_usable_cpus": cores,
        "sample": f"oracle (faithful restatement of c/kmers.c: 2 linear-probe tables at load 0.75, non-rolling encode) "
                  f"on the first {n1} reads ({int(offs[n1]) / 1e6:.1f} Mbases) of the generator, same 2x{n_list} {k}-mer tables, {dt1:.1f} s",
        "all_cores": {"value": round(raten, 6), "unit": "Gbases/s", "cores": cores,
                      "sample": f"first {nall} reads ({int(offs[nall]) / 1e6:.1f} Mbases), reads sharded over {cores} threads, {dtn:.1f} s"},
        "optimised_rolling_all_cores": {"value": round(ratef, 6), "unit": "Gbases/s", "cores": cores,
                                        "sample": f"fairness datum, not the reference's algorithm: rolling canonical k-mers, "
                                                  f"{sample_reads} reads ({int(offs[sample_reads]) / 1e6:.1f} Mbases) over {cores} threads, {dtf:.1f} s"},
        "table_build_s": round(t_tables, 1),
    }
    return base, parity


# ---- count (find-unique-kmers' counting kernel) -----------------------------------------------------
def bench_count(args, np, kmers, lib, check, dev, dist, world, rank):
    """`--path count`: k-mers of synthetic short reads (one haplotype of an implicit random genome
    with substitution errors, generated in HBM) counted into the table in HBM
    (find_unique_kmers.py:62-103's `kmc` step).  A step is one batch of --count-batch-bases through
    tbk_counter_add_device; each rank counts its own reads into its own table (weak scaling)."""
    k, L, genome = args.k, args.count_read_len, args.count_genome
    R = args.count_batch_bases // L
    total = R * L
    err24 = int(args.error_rate * (1 << 24))

    def dalloc(n):
        p = C.c_void_p()
        check(lib.tbk_device_alloc(dev, n, C.byref(p)))
        return p.value

    n_batches = args.warmup + args.steps
    nres = min(4, n_batches)
    res = []
    for b in range(nres):
        d_bases, d_offs = dalloc(total + 64), dalloc((R + 1) * 8)
        res.append((d_bases, d_offs))
    capacity = int(genome * 1.05 + n_batches * total * args.error_rate * k * 1.1) + (1 << 20)
    ctr = kmers.KmerCounter(k, capacity, device=dev)

    def fill(slot, index):
        check(lib.tbk_synth_hap_reads_device(dev, KEY_SEED, genome, 0, READ_SEED + 1, (rank * n_batches + index) * R, R, L, err24,
                                             C.c_void_p(res[slot][0]), C.c_void_p(res[slot][1])))

    # every batch holds new reads (counting the same reads again would only meet known k-mers), so
    # batches are generated outside the clock, `nres` at a time, and counted inside it
    def run(first, n):
        spent = 0.0
        for base in range(first, first + n, nres):
            m = min(nres, first + n - base)
            for s in range(m):
                fill(s, base + s)
            check(lib.tbk_device_sync(dev))
            dist.barrier()
            t0 = time.perf_counter()
            for s in range(m):
                ctr.add_device(res[s][0], res[s][1], R, total)
            check(lib.tbk_device_sync(dev))
            dist.barrier()
            spent += time.perf_counter() - t0
        return spent

    run(0, args.warmup)
    ctr.kernel_timing(True)
    adds_before = C.c_uint64()
    check(lib.tbk_counter_adds_issued(ctr._h, C.byref(adds_before)))
    adds_before = adds_before.value
    elapsed = dist.reduce(run(args.warmup, args.steps), "MAX")
    launches, wins, kernel_ms = ctr.kernel_timing(True)
    bases_all = dist.reduce(args.steps * total, "SUM")
    value = bases_all / elapsed / 1e9
    hist = ctr.histogram()
    st = ctr.stats()
    # algorithmic bytes per window start: 1 read byte + the 8-byte key it is compared with + the
    # 32-bit counter read and written back
    alg = wins * (1 + 8 + 8)
    k_s = kernel_ms * 1e-3
    out = {
        "metric": "Gbases/sec counted (find-unique-kmers counting step, k=%d)" % k, "value": round(value, 3), "unit": "Gbases/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"k-mer counting: {L} b synthetic reads of a {genome / 1e6:.0f} Mb genome with {args.error_rate:g} errors per base, "
                               f"{R} reads ({total / 1e9:.2f} Gbases) per step per GPU resident in HBM, canonical {k}-mers into a counting table in HBM",
                   "k": k, "read_len": L, "reads_per_step": R, "genome": genome, "table_bytes_per_gpu": st["table_bytes"],
                   "table_load": round(int(hist[0]) / st["n_slots"], 3), "distinct_kmers": int(hist[0])},
        "roofline": {"bound": "hbm", "achieved": round(alg / k_s / 1e9, 1) if k_s > 0 else 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(alg / k_s / 1e9 / HBM_PEAK_GBPS, 4) if k_s > 0 else 0.0, "traffic": None,
                     "kernel": "tbk_count_kernel", "kernel_ms_avg": round(kernel_ms / max(1, launches), 4), "launches": int(launches),
                     "alg_bytes_per_window": 17, "windows": int(wins),
                     "kernel_only_gbases_per_s": round(args.steps * total / k_s / 1e9, 2) if k_s > 0 else None,
                     "window_starts_per_s": round(wins / k_s / 1e9, 2) if k_s > 0 else None},
        "device": __import__("trio_binning_amd")._lib.device_name(dev),
    }
    # a step is several launches (the stream is counted in pieces of a quarter of the table's slots): kernel time per STEP
    # beside the step's wall time says what the clock holds besides the counting kernel (the separating kernel, two small
    # synchronising copies per piece)
    out["roofline"]["launches_per_step"] = round(launches / max(1, args.steps), 2)
    out["roofline"]["kernel_ms_per_step"] = round(kernel_ms / max(1, args.steps), 3)
    out["roofline"]["step_ms_outside_the_kernel"] = round(elapsed / args.steps * 1e3 - kernel_ms / max(1, args.steps), 3)
    if rank == 0:
        # The yardstick: the chip's rate for the adds the kernel really issues - 64-bit atomic adds that count two
        # neighbouring 32-bit counters at once, one or two per line visited - measured by tbk_calib_atomics64 over the
        # same footprint; the kernel's adds are counted on the device (tbk_counter_adds_issued).  (Round 3 priced
        # window starts against the rate of 32-bit adds and read 1.43: merged adds count two windows.)
        adds = C.c_uint64()
        check(lib.tbk_counter_adds_issued(ctr._h, C.byref(adds)))
        adds_timed = adds.value - adds_before
        run_len = max(1, min(4, int(round(adds_timed / max(1.0, wins * 0.29)))))  # adds per line visited (a window changes lines with density ~0.29)
        aps = C.c_double()
        check(lib.tbk_calib_atomics64(dev, min(st["table_bytes"], 40 << 30), run_len, 3, C.byref(aps)))
        aps32 = C.c_double()
        check(lib.tbk_calib_atomics(dev, min(st["table_bytes"], 40 << 30), 3, 3, C.byref(aps32)))
        out["roofline"]["atomic_adds_issued"] = int(adds_timed)
        out["roofline"]["atomic_adds_per_window"] = round(adds_timed / max(1, wins), 4)
        out["roofline"]["atomic_adds_Gps"] = round(adds_timed / k_s / 1e9, 2) if k_s > 0 else None
        out["roofline"]["atomic_adds_ceiling_Gps"] = round(aps.value / 1e9, 2)
        out["roofline"]["atomic_ceiling"] = f"64-bit adds, {run_len} per random line (tbk_calib_atomics64)"
        out["roofline"]["atomic_frac"] = round(adds_timed / k_s / aps.value, 3) if k_s > 0 else None
        out["roofline"]["atomic_adds32_ceiling_Gps"] = round(aps32.value / 1e9, 2)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle

        orc = oracle.load()
        n_cpu = min(R, max(1000, int(args.cpu_seconds * 25e6 // L)))
        h_bases = np.empty(n_cpu * L, dtype=np.uint8)
        h_offs = np.arange(n_cpu + 1, dtype=np.uint64) * np.uint64(L)
        check(lib.tbk_memcpy_d2h(dev, h_bases.ctypes.data, C.c_void_p(res[0][0]), h_bases.size))
        t = time.perf_counter()
        cpu_hist = orc.kmer_histogram(h_bases, h_offs, k, n_cpu * L)
        cpu_s = time.perf_counter() - t
        with kmers.KmerCounter(k, n_cpu * L, device=dev) as small:
            small.add(h_bases, h_offs)
            same = bool(np.array_equal(small.histogram(), cpu_hist))
        out["cpu_baseline"] = {"value": round(n_cpu * L / cpu_s / 1e9, 6), "unit": "Gbases/s", "cores": 1, "kind": "port",
                               "sample": f"oracle C counter (one open-addressing table, rolling canonical k-mers; KMC itself is not in the reference checkout) "
                                         f"on {n_cpu} reads ({n_cpu * L / 1e6:.0f} Mbases) of the last batch, {cpu_s:.1f} s"}
        out["parity"] = {"gpu_histogram_equals_cpu": same, "reads_checked": int(n_cpu), "distinct_kmers_checked": int(cpu_hist[0])}
    ctr.close()
    return out


if __name__ == "__main__":
    main()
