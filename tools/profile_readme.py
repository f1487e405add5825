#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_profile.sh run from gpurun_out/ into profiles/<round>/ and write that
directory's README.md from the JSON lines themselves (no number is typed by hand).

    python tools/profile_readme.py r04
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
KEEP = ["bench_default.json", "bench_default_key_layout.json", "bench_haplotypes.json", "bench_2ranks_shared_device.json", "bench_2ranks_torchrun.json", "bench_8ranks_shared_device.json", "bench_strong.json", "bench_count.json", "bench_3rings.json", "bench_c5_uniform.json", "bench_c5_haplotypes.json",
        "kernel_stats.csv", "count_kernel_stats.csv", "kernel_trace_by_launch_size.json", "pmc_summary_uniform.json", "pmc_summary_haplotypes.json",
        "reader_hifi.json", "cli_configs1.json", "cli_configs1_haplotypes.json", "cli_configs1_one_small_disk.json", "cli_lists_configs2.json", "cli_gz_input.json", "cli_gz_configs1.json", "calib_ceilings.json",
        "bench_c5_uniform.json", "bench_c5_uniform_key_layout.json", "bench_c5_haplotypes.json", "bench_c5_lognormal_uniform.json", "bench_c5_lognormal_haplotypes.json", "ab_full_unroll.log", "bench_count.json", "bench_3rings.json"]
for name in KEEP:
    p = os.path.join(src, name)
    if os.path.isfile(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, name))
for name in ("pmc_traffic.json", "pmc_traffic_haplotypes.json"):  # bench.py replays these (roofline.traffic)
    p = os.path.join(src, name)
    if os.path.isfile(p):
        shutil.copy(p, os.path.join(ROOT, "profiles", name))
        if "--restamp" in sys.argv:
            # the passes ran on this tree's kernels (same sources, same compiler: the same machine code) but were stamped by an
            # older profile_summary.py with the hash of the source text: put the machine-code hash in its place
            sys.path.insert(0, ROOT)
            from bench import kernel_fingerprint

            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            t.pop("kernel_source_sha256", None)
            t["kernel_sha256"] = kernel_fingerprint()
            json.dump(t, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)


# the bench line printed by the run that rocprofv3 traced: its HIP-event durations belong beside the trace's
try:
    lines = [l for l in open(os.path.join(src, "prof_trace.log")) if l.startswith('{"metric"')]
    if lines:
        open(os.path.join(dst, "bench_under_rocprofv3.json"), "w").write(lines[-1])
except OSError:
    pass


# the ceilings of profiles/calibration.json (tools/calib_ceilings.py: the gather in the probe kernels' shape, the tuned streaming read)
try:
    _cal = json.load(open(os.path.join(ROOT, "profiles", "calibration.json")))
    STREAM_TBPS = _cal.get("stream_tuned_GBps", _cal["guide_stream_GBps"]) / 1e3
    _rows = _cal.get("random_lines_tuned_Glines_per_s") or {}
    GATHER_GLPS = _rows.get("33GB", {}).get("best") or 47.75
except Exception:
    STREAM_TBPS, GATHER_GLPS = 6.29, 47.75


def load(name):
    try:
        return json.load(open(os.path.join(dst, name)))
    except Exception:
        return None


# kernel_stats.csv (rocprofv3 --stats) averages over EVERY launch of a kernel - sliced first batches and the small parity
# launches included; the companion says what the full-size launches took, per kernel, so that the two files agree at a glance
try:
    tr = json.load(open(os.path.join(dst, "kernel_trace_by_launch_size.json")))
    biggest = {}
    for t in tr:
        if t["kernel"] not in biggest or t["grid"] > biggest[t["kernel"]]["grid"]:
            biggest[t["kernel"]] = t
    with open(os.path.join(dst, "kernel_stats_full_size_launches.csv"), "w") as fh:
        fh.write('"Name","GridSize","Calls","AverageMs","MinMs","MaxMs"\n')
        for k, t in sorted(biggest.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
            fh.write('"%s",%d,%d,%.4f,%.4f,%.4f\n' % (k, t["grid"], t["launches"], t["avg_ms"], t["min_ms"], t["max_ms"]))
except Exception:
    pass

u, h = load("bench_default.json"), load("bench_haplotypes.json")
pu, ph = load("pmc_summary_uniform.json") or {}, load("pmc_summary_haplotypes.json") or {}
trace = load("kernel_trace_by_launch_size.json") or []
two, tr2, strong, count = load("bench_2ranks_shared_device.json"), load("bench_2ranks_torchrun.json"), load("bench_strong.json"), load("bench_count.json")
cli, reader = load("cli_configs1.json"), load("reader_hifi.json")


def g(d, *path, default="-"):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def col(b, p):
    if not b:
        return ["-"] * 20
    r, c = b["roofline"], b["config"]
    w = r["windows_per_launch"]
    lines = p.get("TCC_MISS_sum", 0) / w if w and p else None
    hbm = p.get("hbm_bytes_per_launch_from_FETCH_SIZE")
    ms = r["kernel_ms_avg"]
    insts = lambda k: round(p[k] / w, 2) if p and k in p and w else "-"
    return [
        f"{b['value']} ({b['ms_per_step']} ms per {c['bases_per_step_per_rank'] / 1e9:.2f}-Gbase step)",
        f"{g(b, 'kernel_resident', 'gbases_per_s')}",
        f"{c['bucket_select']}, load {c['table_load']}, {c['line_layout'].split(':')[0]} layout, {c['table_bytes_per_gpu'] / 1e9:.0f} GB ({c['table_bytes_per_key']} B per key)",
        f"{ms} ({r['launches']} launches); whole probe {r['whole_probe_ms_avg']}",
        f"{r['alg_bytes_per_launch'] / 1e9:.1f} / {r['achieved']} / **{r['frac']}** (two-probe reading, P = 2: {r.get('frac_P2_two_probe_reading')})",
        f"{hbm / 1e9:.1f} / {p.get('hbm_bytes_per_launch_from_TCC_MISS', 0) / 1e9:.1f}" if hbm else "-",
        f"{hbm / w:.1f} / {lines:.4f}" if hbm and lines else "-",
        f"{hbm / ms / 1e9:.2f} / {hbm / ms / 1e9 / STREAM_TBPS:.2f}" if hbm else "-",
        (f"{r.get('random_lines_Gps')} / {r.get('random_line_frac')} / **{r.get('random_line_frac_same_table')}** (this run's table: {g(r, 'random_line_ceiling_same_table_Gps', 'best')} G/s)"
         if r.get("random_lines_Gps") else "-"),
        f"{insts('SQ_INSTS_VALU')} / {insts('SQ_INSTS_SALU')} / {insts('SQ_INSTS_VMEM_RD')}",
        f"{p['SQ_WAIT_ANY'] / p['SQ_WAVE_CYCLES']:.2f}" if p.get("SQ_WAVE_CYCLES") else "-",
        f"{g(b, 'parity', 'gpu_equals_cpu')} on {g(b, 'parity', 'reads_checked_against_the_oracle')} reads; transfers agree: {g(b, 'parity', 'packed_and_ascii_transfers_agree')}",
        f"{g(b, 'cpu_baseline', 'value', default=0) * 1e3:.2f} / {g(b, 'cpu_baseline', 'all_cores', 'value', default=0) * 1e3:.1f} / {g(b, 'cpu_baseline', 'optimised_rolling_all_cores', 'value', default=0) * 1e3:.1f}",
        f"{g(b, 'pipeline_variants', 'ascii_in_packed_by_feeder', 'gbases_per_s')} / {g(b, 'pipeline_variants', 'ascii_over_pcie', 'gbases_per_s')}",
    ]


rows = ["`value`: host-fed classify stage (Gbases/s)", "`kernel_resident` (Gbases/s)", "table", "single-read probe kernel, HIP events inside the timed region (ms per launch)",
        "its algorithmic bytes per launch (GB, P = 1: 9 B per window) / `roofline.achieved` (GB/s) / `frac`", "its HBM bytes per launch: FETCH_SIZE x 1024 x 2 / TCC_MISS x 128 B (GB)",
        "bytes per window / 128-B lines per window", f"HBM traffic rate (TB/s) / of the {STREAM_TBPS:.2f} TB/s a tuned streaming read reaches (profiles/calibration.json)", f"random 128-B lines (G/s; the bench line's own counter pass, the read stream's bytes taken off) / of the {GATHER_GLPS:.2f} G/s a gather in the kernels' own shape reached on the calibration's box / of the same gather over this run's own table, in this run",
        "VALU / SALU / VMEM-read instructions per window", "SQ_WAIT_ANY / SQ_WAVE_CYCLES", "parity in the run (GPU counts == oracle)",
        "CPU baseline, oracle: 1 thread / 16 CPUs / rolling, 16 CPUs (Mbases/s)", "same stage fed with ASCII batches: packed by the feeder / ASCII over PCIe (Gbases/s)"]
cu, ch = col(u, pu), col(h, ph)
full = [t for t in trace if "tbk_probe_kernel" in t["kernel"] or "tbk_probe_entry_kernel" in t["kernel"]][:8]

out = [f"# Reference profile of round {rnd[1:].lstrip('0')} ({rnd})", "",
       "Made by `tools/gpu_profile.sh` on one MI355X box (ROCm 7.2), condensed by `tools/profile_summary.py`, and this file by",
       "`tools/profile_readme.py` from the JSON lines beside it - no number here is typed by hand.  Box-to-box spread of the pool: +-4 %.", "",
       "    python bench.py                                  -> bench_default.json (host-fed value, kernel_resident, pipeline variants, cpu_baseline, parity)",
       "    python bench.py --lists haplotypes               -> bench_haplotypes.json (lists shaped like real find-unique-kmers output)",
       "    python bench.py --gpus 2 --share-device          -> bench_2ranks_shared_device.json (two ranks on the one GPU: plumbing of the N > 1 line)",
       "    python -m torch.distributed.run ... bench.py --gpus 2 --share-device   -> bench_2ranks_torchrun.json (the driver's launcher)",
       "    python bench.py --gpus 8 --share-device --kmers-per-list 30000000 ...  -> bench_8ranks_shared_device.json (eight ranks, eight tables, one GPU)",
       "    python bench.py --scaling strong --strong-reads 6000000 --steps 2      -> bench_strong.json (BASELINE configs[2]: the 90 Gbp set, host-fed, one rank)",
       "    python bench.py --path count                     -> bench_count.json",
       "    rocprofv3 --kernel-trace --stats -- python3 bench.py                   -> kernel_stats.csv, kernel_trace_by_launch_size.json",
       "    rocprofv3 --kernel-trace --pmc <one set per run> -- python3 bench.py [--lists haplotypes] --steps 4 --warmup 1 ...   -> pmc_summary_<lists>.json", "",
       "Workload: bench.py defaults = k = 21, 2 x 3e8 keys, 262144 x 15 kb reads (3.932 Gbases) per step, 20 steps per timed region.", "",
       "| | uniform lists (BASELINE) | haplotype-shaped lists |", "|---|---|---|"]
out += [f"| {r} | {a} | {b} |" for r, a, b in zip(rows, cu, ch)]
rl = g(u, "realistic_lists", default=None)
if isinstance(rl, dict):
    out += ["", f"`realistic_lists` inside the default line (the same run, haplotype-shaped lists): value {rl['value']} Gbases/s host-fed, kernel_resident {rl['kernel_resident']}, "
            f"single-read kernel {rl['kernel_ms_avg']} ms, frac {rl['frac']} (P = 2: {rl['frac_P2_two_probe_reading']}), random_line_frac {rl.get('random_line_frac')}, "
            f"{rl['table_bytes_per_key']} B of table per key ({rl['line_layout'].split(':')[0]}), parity gpu_equals_cpu = {g(rl, 'parity', 'gpu_equals_cpu')} on {g(rl, 'parity', 'reads_checked_against_the_oracle')} reads."]
out += ["", "rocprofv3 `--kernel-trace` of `python3 bench.py`, per kernel and launch size (`kernel_stats.csv` averages over every launch of a kernel,",
        "the two small parity launches included; the roofline's duration is that of the full-size launches):", "",
        "| kernel | grid | launches | avg ms | min | max |", "|---|---|---|---|---|---|"]
out += [f"| `{t['kernel']}` | {t['grid']} | {t['launches']} | {t['avg_ms']} | {t['min_ms']} | {t['max_ms']} |" for t in full]
prof = load("bench_under_rocprofv3.json")
if prof:
    out += ["", f"The traced process's own bench line (`bench_under_rocprofv3.json`): single-read kernel by HIP events {prof['roofline']['kernel_ms_avg']} ms in the host-fed region, "
            f"{g(prof, 'kernel_resident', 'single_read_kernel_ms_avg')} ms in the resident region - the trace's average over the same process's full-size launches is the figure above "
            f"(value {prof['value']}, kernel_resident {g(prof, 'kernel_resident', 'gbases_per_s')} Gbases/s under the profiler; separate processes on one box differ by up to 5 %)."]
if two:
    out += ["", f"Two ranks on the one device (`--share-device`; a plumbing run, not a scaling result): value {two['value']} Gbases/s, kernel_resident {g(two, 'kernel_resident', 'gbases_per_s')}, "
            f"parity all_ranks_equal = {g(two, 'parity', 'all_ranks_equal')}, gpu_equals_cpu = {g(two, 'parity', 'gpu_equals_cpu')}, devices {json.dumps(two.get('devices'))}."]
eight = load("bench_8ranks_shared_device.json")
if eight:
    out += [f"Eight ranks on the one device (2 x {g(eight, 'config', 'kmers_per_list')} keys so that eight tables fit; plumbing, not scaling): value {eight['value']} Gbases/s, "
            f"all_ranks_equal = {g(eight, 'parity', 'all_ranks_equal')}, gpu_equals_cpu = {g(eight, 'parity', 'gpu_equals_cpu')}; per rank (NUMA node, CPUs bound to, host threads): "
            f"{[(d.get('numa_node'), d.get('cpus_bound_to'), d.get('host_threads')) for d in eight.get('devices', [])]}."]
if tr2:
    out += [f"Under `python -m torch.distributed.run`: value {tr2['value']}, parity all_ranks_equal = {g(tr2, 'parity', 'all_ranks_equal')}."]
keyl = load("bench_default_key_layout.json")
if keyl:
    out += [f"The uniform lists in the key layout (`TBK_SHORT=0`: what they got before short keys), same box, 5 s regions: value {keyl['value']} Gbases/s host-fed, "
            f"kernel_resident {g(keyl, 'kernel_resident', 'gbases_per_s')}, single-read kernel {g(keyl, 'roofline', 'kernel_ms_avg')} ms, frac {g(keyl, 'roofline', 'frac')}, "
            f"{g(keyl, 'config', 'table_bytes_per_gpu', default=0) / 1e9:.0f} GB ({g(keyl, 'config', 'table_bytes_per_key')} B per key); "
            f"short keys in the default line above: {g(u, 'value')} / {g(u, 'kernel_resident', 'gbases_per_s')}, {g(u, 'roofline', 'kernel_ms_avg')} ms, "
            f"{g(u, 'config', 'table_bytes_per_gpu', default=0) / 1e9:.0f} GB ({g(u, 'config', 'table_bytes_per_key')} B per key).", ""]
rings3 = load("bench_3rings.json")
if rings3:
    out += [f"Three feeder threads + rings on the one device, sharing its table (`--rings 3`: the pipeline's dealing from one queue, as it would run over three GPUs): value {rings3['value']} Gbases/s "
            f"(one ring, same script: {g(u, 'value')}; the rings of one device share its three streams), "
            f"batches per ring {g(rings3, 'config', 'batches_per_ring')}, parity {g(rings3, 'parity', 'all_ranks_equal')}."]
for nm, label in (("bench_c5_uniform.json", "uniform"), ("bench_c5_haplotypes.json", "haplotype-shaped")):
    c5 = load(nm)
    if c5:
        cc = c5["config"]
        out += [f"BASELINE configs[4]'s table and read shape on one GPU (k = 31, 2 x {cc['kmers_per_list']} keys, {cc['read_len']} b reads, {label} lists): value {c5['value']} Gbases/s host-fed, "
                f"kernel_resident {g(c5, 'kernel_resident', 'gbases_per_s')}; {cc['bucket_select']}, load {cc['table_load']}, {cc['table_bytes_per_gpu'] / 1e9:.0f} GB table, {cc['line_layout'].split(':')[0]} layout; "
                f"transfers agree: {g(c5, 'parity', 'packed_and_ascii_transfers_agree')} (oracle parity at this scale: tests/test_gpu_scale.py)."]
if strong:
    out += [f"BASELINE configs[2] literally (`--scaling strong`, one rank: {g(strong, 'config', 'workload')[:110]}...): value {strong['value']} Gbases/s, {strong['ms_per_step']} ms per pass over the set, "
            f"{g(strong, 'config', 'launches_per_step')} host-fed batches per pass, gpu_equals_cpu = {g(strong, 'parity', 'gpu_equals_cpu')}."]
if count:
    cr = count["roofline"]
    out += [f"`--path count`: {count['value']} Gbases/s counted ({count['ms_per_step']} ms per step = {cr.get('launches_per_step')} launches, {cr.get('kernel_ms_per_step')} ms of them in the counting kernel); "
            f"{cr.get('atomic_adds_per_window')} 64-bit atomic adds per window start, {cr.get('atomic_adds_Gps')} G adds/s against the chip's {cr.get('atomic_adds_ceiling_Gps')} ({cr.get('atomic_ceiling')}): "
            f"atomic_frac {cr.get('atomic_frac')}; histogram parity {g(count, 'parity', 'gpu_histogram_equals_cpu')}."]
if cli:
    out += ["", f"End to end (`tools/measure_e2e.py`, {cli['config']}; {cli['fastq_GB']} GB of FASTQ, {cli['lists_GB']} GB of list text; page cache {cli['page_cache']}; {cli['host_usable_cpus']} usable CPUs):", "",
            "| lists | both lists (s) | M lines/s | paired table build (s) |", "|---|---|---|---|"]
    out += [f"| {k} | {v['both_lists_s']} | {v['Mlines_per_s']} | {v['paired_table_build_s']} |" for k, v in cli.get("lists", {}).items()]
    out += ["", "| run | wall (s) | Gbases/s, wall | loop (s) | reader busy (s) | writer busy (s) | waiting for the GPU (s) | before the loop (s) | output (GB) |", "|---|---|---|---|---|---|---|---|---|"]
    def runs(c):
        rows = []
        for k, v in c.items():
            if isinstance(v, dict) and "wall_s" in v:
                st = v["stages"]
                rows += [f"| {k} | {v['wall_s']} | {v['gbases_per_s_wall']} | {st.get('loop_s')} | {st.get('read_s')} | {st.get('write_s')} | {st.get('gpu_wait_s')} | {v['before_the_loop_s']} | {v['out_GB']} |"]
        return rows
    out += runs(cli)
    out += ["", f"(inputs on {cli.get('inputs_on')}, bins on {cli.get('outputs_on')}; inputs sync'ed in {cli.get('inputs_synced_s')} s before the runs.)"]
    big = load("cli_lists_configs2.json")
    if big:
        out += ["", f"List loading at BASELINE configs[2] scale ({big['config']}; {big['lists_GB']} GB of list text):", "",
                "| lists | both lists (s) | M lines/s | paired table build (s) |", "|---|---|---|---|"]
        out += [f"| {k} | {v['both_lists_s']} | {v['Mlines_per_s']} | {v['paired_table_build_s']} |" for k, v in big.get("lists", {}).items()]
    small = load("cli_configs1_one_small_disk.json")
    if small:
        out += ["", f"The same with inputs and bins on the one {g(small, 'outputs_on', 'total_GB')} GB file system of the box (inputs + key caches + bins = 69 GB of it: ext4 runs low on free space against its dirty data and",
                "stops the writer for about 2 s near the end of a run - `e2e_writer_stall.log`):", "",
                "| run | wall (s) | Gbases/s, wall | loop (s) | reader busy (s) | writer busy (s) | waiting for the GPU (s) | before the loop (s) | output (GB) |", "|---|---|---|---|---|---|---|---|---|"]
        out += runs(small)
        wt = [v.get("write_timing") for v in small.values() if isinstance(v, dict) and v.get("write_timing")]
        if wt:
            out += ["", "Writer, " + wt[0][len("tbk-write-timing "):] + "."]
if reader:
    out += ["", f"Reader alone (`tools/measure_reader.py --qual hifi`, {reader['text_GB']} GB of FASTQ text, GB/s of text): plain first pass {g(reader, 'plain_first_pass', 'text_GB_per_s')}, "
            f"plain warm {g(reader, 'plain', 'text_GB_per_s')}, gzip {g(reader, 'gzip', 'text_GB_per_s')}, bgzf {g(reader, 'bgzf', 'text_GB_per_s')}."]
out += ["", "Experiment logs of the round (same-box A/B runs; `EXPERIMENTS.md` reads them): `ab_entry_layout.log` (entry layout on / off, entries per bucket, uniform lists forced",
        "into entries, spans of six and seven m-mers), `ab_h2d.log` (one / two H2D streams, copy stream priority, blit kernels), `calib_shape.json` (random-line rate by access shape),",
        "`valu_rates.log` (instruction throughput), `gate_scatter.log` (partition-then-probe gate), `gate_write_direct.log` (O_DIRECT bins gate),",
        "`ab_wide_entries.log` (wide entries: m-mer length, loads), `ab_span3.log` (mod-sampling over 2w and 3w t-mer positions per span), `ab_loads_span3.log` (keys per line /",
        "entries per bucket under 3w sampling), `ab_wide_span3.log` (wide entries over 24 positions: dropped), `ab_short_drain.log` (short keys: when the back queue drains),",
        "`bench_haplotypes_repeats_*.json` (lists from a genome with repeat families, entry layout and key layouts).", ""]
open(os.path.join(dst, "README.md"), "w").write("\n".join(out))
print("\n".join(out[:40]))
