#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh run (gpurun_out/) into the small files kept under profiles/rNN_*:
kernel_stats.csv (rocprofv3 --stats), pmc_summary_<lists>.json (per-launch means of every PMC pass for
the single-read probe kernel, the dominant one) and pmc_traffic[_haplotypes].json (HBM bytes per window, which bench.py scales to
its own launch size for `roofline.traffic`).

HBM bytes from the counters as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE is reported in KiB and tallies the L2's 128-byte memory-side requests at 64 bytes, so
bytes = FETCH_SIZE x 1024 x 2; cross-check: TCC_MISS_sum x 128 B (separate pass).  WRITE_SIZE x 1024."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import counter_means, is_single_read_probe, kernel_fingerprint  # noqa: E402


def probe_means(pattern):
    """Per-launch means over the FULL-SIZE launches of the single-read kernel (bench.counter_means)."""
    means, meta = counter_means(sorted(glob.glob(os.path.join(out_dir, pattern, "*", "*_counter_collection.csv"))))
    if meta:
        meta["VGPR_note"] = ("rocprofv3's VGPR column is not the allocation: the code object's .vgpr_count, LDS bytes and the waves per SIMD they allow are in the bench line "
                             "(roofline.kernel_resources) and below (kernel_resources)")
    return means, meta


def trace_summary(name, dst):
    """rocprofv3 --kernel-trace of a bench run, per kernel and launch size: the stats CSV averages over every
    launch of a kernel, small parity and variant batches included; the roofline's duration is that of the
    full-size launches."""
    hits = glob.glob(os.path.join(out_dir, name, "*", "*_kernel_trace.csv"))
    if not hits:
        return
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(hits[0])):
        if "tbk_" in r["Kernel_Name"]:
            groups[(r["Kernel_Name"].split("(")[0][:70], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out = [{"kernel": k, "grid": g, "launches": len(v), "avg_ms": round(sum(v) / len(v), 4), "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)}
           for (k, g), v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))]
    json.dump(out, open(os.path.join(out_dir, dst), "w"), indent=1)
    print(dst, json.dumps(out[:6]))


for lists in ("uniform", "haplotypes"):
    summary = {}
    for d in sorted(glob.glob(os.path.join(out_dir, f"pmc_{lists}_*"))):
        if not os.path.isdir(d):
            continue
        means, meta = probe_means(os.path.basename(d))
        summary.update({k: round(v) for k, v in means.items()})
        summary.update({k: v for k, v in meta.items() if v})
    if not summary:
        continue
    bench = None
    bfile = os.path.join(out_dir, "bench_default.json" if lists == "uniform" else "bench_haplotypes.json")
    try:
        bench = json.loads(open(bfile).read())
    except Exception:
        pass
    # the PMC passes run bench.py with its default batch: windows per launch from the bench line of the same shape
    if bench and bench.get("roofline", {}).get("kernel_resources"):
        summary["kernel_resources"] = bench["roofline"]["kernel_resources"]
    if bench and "FETCH_SIZE" in summary:
        windows = bench["roofline"]["windows_per_launch"]
        hbm = summary["FETCH_SIZE"] * 1024 * 2
        summary["hbm_bytes_per_launch_from_FETCH_SIZE"] = hbm
        summary["hbm_bytes_per_launch_from_TCC_MISS"] = summary.get("TCC_MISS_sum", 0) * 128
        summary["windows_per_launch"] = windows
        summary["lines_per_window"] = round(summary.get("TCC_MISS_sum", 0) / windows, 4)
        cfg = bench["config"]
        traffic = {
            "kernel": "tbk_probe_kernel<single-read>",
            "kernel_sha256": kernel_fingerprint(),  # bench.py replays this record only on these very kernels (hash of their machine code)
            "note": "HBM bytes of one launch of the single-read probe kernel (tbk_probe_kernel<..., MULTI = false>), from rocprofv3 PMC passes run separately from the timed bench "
                    "(tools/gpu_profile.sh): FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM), "
                    "cross-checked by TCC_MISS_sum x 128 B; divided by the launch's window starts so that bench.py can scale it to its own launch size.",
            "read_len": cfg["read_len"], "kmers_per_list": cfg["kmers_per_list"], "k": cfg["k"], "bucket_select": cfg["bucket_select"],
            "table_load": cfg["table_load"], "lists": lists, "front_layout": str(cfg.get("line_layout", "")).startswith("front"), "entry_layout": str(cfg.get("line_layout", "")).startswith("entries"), "short_keys": str(cfg.get("line_layout", "")).startswith("short"), "windows_per_launch": windows,
            "hbm_bytes_per_launch": hbm, "hbm_bytes_per_window": hbm / windows,
            "fetch_size_kib_raw": summary["FETCH_SIZE"], "write_size_kib_raw": summary.get("WRITE_SIZE"),
            "tcc_miss": summary.get("TCC_MISS_sum"), "tcc_hit": summary.get("TCC_HIT_sum"),
        }
        json.dump(traffic, open(os.path.join(out_dir, "pmc_traffic.json" if lists == "uniform" else "pmc_traffic_haplotypes.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(out_dir, f"pmc_summary_{lists}.json"), "w"), indent=1)
    print(lists, json.dumps(summary))

trace_summary("prof_trace", "kernel_trace_by_launch_size.json")
for name, dst in (("prof_trace", "kernel_stats.csv"), ("prof_trace_count", "count_kernel_stats.csv")):
    hits = glob.glob(os.path.join(out_dir, name, "*", "*_kernel_stats.csv"))
    if hits:
        shutil.copy(hits[0], os.path.join(out_dir, dst))
        print(dst, open(hits[0]).read()[:900])
