"""bench.py as the driver runs it, at test size: one JSON line with the contract's keys, and `roofline.traffic` taken from
counters of the run itself (the child pass under rocprofv3 --pmc FETCH_SIZE), not replayed from profiles/."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = ["--kmers-per-list", "3000000", "--reads-per-step", "8192", "--steps", "3", "--warmup", "1", "--min-timed-s", "0",
         "--no-cpu-baseline", "--no-streaming", "--no-realistic", "--no-strong-leg", "--parity-reads", "256"]


def run_bench(extra, base=None):
    env = dict(os.environ, TBK_SKIP_BUILD="1")
    for key in [k for k in env if k.startswith(("ROCPROF", "ROCP_"))]:
        env.pop(key)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + (SMALL if base is None else base) + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    out = run_bench(["--live-pmc", "off"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "parity"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "Gbases/s" and out["value"] > 0
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "_pricing" not in rf and "switched off" in rf["live_counter_pass"]
    assert out["parity"]["all_ranks_equal"] and out["parity"]["packed_and_ascii_transfers_agree"]
    assert out["sweep"]["ok"]


@pytest.mark.skipif(shutil.which("rocprofv3") is None and not os.path.isfile("/opt/rocm/bin/rocprofv3"), reason="no rocprofv3")
def test_traffic_is_measured_in_the_run():
    out = run_bench(["--live-pmc", "on", "--no-sweep"])
    rf = out["roofline"]
    assert "live_counter_pass" not in rf, rf.get("live_counter_pass")
    assert rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
    # every window's read byte comes from HBM at least once, and a launch cannot fetch more lines than it has windows (x 128 B, + the reads)
    assert rf["windows_per_launch"] * 0.25 <= rf["traffic"] <= rf["windows_per_launch"] * 130
    assert rf["traffic_over_algorithmic"] == round(rf["traffic"] / rf["alg_bytes_per_launch"], 3)
    # the line rate is priced against a gather over this run's own table as well
    same = rf["random_line_ceiling_same_table_Gps"]
    assert same["best"] == max(v for name, v in same.items() if name != "best") > 0
    assert rf["random_line_frac_same_table"] == round(rf["random_lines_Gps"] / same["best"], 3) or abs(rf["random_line_frac_same_table"] - rf["random_lines_Gps"] / same["best"]) < 2e-3


@pytest.mark.skipif(shutil.which("rocprofv3") is None and not os.path.isfile("/opt/rocm/bin/rocprofv3"), reason="no rocprofv3")
def test_sub_records_of_the_default_line():
    """The two sub-records the default N = 1 run adds, at test size: `strong_90gbp` (BASELINE configs[2]'s literal set: one pass
    per step over every distinct batch of a fixed read set, with its own parity against the oracle) and `realistic_lists`
    (haplotype-shaped lists) with its `traffic` measured by a counter pass of its own - the headline's treatment."""
    base = [a for a in SMALL if a not in ("--no-realistic", "--no-strong-leg", "--no-cpu-baseline", "--no-streaming")]
    out = run_bench(["--live-pmc", "on", "--no-sweep", "--strong-reads", "20000", "--strong-leg-timed-s", "0", "--realistic-timed-s", "0", "--cpu-seconds", "0.5"], base)
    st = out["strong_90gbp"]
    assert st["reads"] == 20000 and st["distinct_batches"] == 3 and st["bases"] == 20000 * 15000 and st["value"] > 0
    assert st["parity"]["all_ranks_equal"] and st["parity"]["gpu_equals_cpu"], st["parity"]
    assert sum(st["bins_of_one_pass"].values()) == 20000
    rl = out["realistic_lists"]
    assert rl["traffic_source"].startswith("measured in this run"), rl["traffic_source"]
    assert rl["traffic"] > 0 and rl["parity"]["gpu_equals_cpu"]
    assert out["roofline"]["traffic_source"].startswith("measured in this run")
    enc = out["bins_gzip_encoder"]   # the default line's record of the GPU gzip encoder behind the bins
    assert enc["text_GB_per_s"] > 1 and 0.2 < enc["ratio"] < 0.6 and enc["roofline"]["bound"] == "pcie"
    inf = out["input_bgzf_inflater"]   # and of the GPU inflater in front of the reader (every block's CRC-32 checked in the timed region)
    assert inf["kernels_only_text_GB_per_s"] > 1 and inf["ring_text_GB_per_s"] > 0.5 and 0.2 < inf["ratio"] < 0.6 and inf["blocks"] > 10
