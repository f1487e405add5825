#!/bin/bash
# round 3, batch C: repeatability of the 5-wave single-read kernel, denser tables in the front layout, haplotype lists
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
one() {  # label, env..., -- bench flags
  python - "$@" <<'PY'
import json, os, subprocess, sys
label = sys.argv[1]; rest = sys.argv[2:]; i = rest.index("--"); envs, flags = rest[:i], rest[i + 1:]
env = dict(os.environ); env.update(e.split("=", 1) for e in envs)
p = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-streaming"] + flags, env=env, capture_output=True, text=True, timeout=900)
line = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
if not line:
    print(label, "FAILED", p.stderr[-400:]); sys.exit(0)
d = json.loads(line[-1]); c = d["config"]; r = d["roofline"]
print(label, "|", d["value"], "Gb/s  kernel_ms", r["kernel_ms_avg"], "regions", d["region_s_min_median_max"], "| load", c["table_load"], c["bucket_select"], "|", c["line_layout"][:11], "| table GB", round(c["table_bytes_per_gpu"] / 1e9, 1),
      "| builds", c["layout_builds"], "past", c["keys_past_their_half"], "behind", c["keys_behind_front"], flush=True)
for l in p.stderr.splitlines():
    if l.startswith("tbk-counters"): print("   ", l, flush=True)
PY
}
{
for round in 1 2 3 4; do
  one "uniform a_w5" TBK_LIBRARY=$V/a_w5.so --
  one "uniform b_w4" TBK_LIBRARY=$V/b_w4.so --
done
one "uniform front .12 w5" TBK_LIBRARY=$V/a_w5.so TBK_FRONT=1 TBK_TABLE_LOAD=0.12 --
one "uniform front .16 w5" TBK_LIBRARY=$V/a_w5.so TBK_FRONT=1 TBK_TABLE_LOAD=0.16 --
one "uniform front .06 w5" TBK_LIBRARY=$V/a_w5.so TBK_FRONT=1 TBK_TABLE_LOAD=0.06 --
for round in 1 2; do
one "hap front,ms,.04 w5" TBK_LIBRARY=$V/a_w5.so TBK_FRONT=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
one "hap front,ms,.08 w5" TBK_LIBRARY=$V/a_w5.so TBK_FRONT=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
one "hap front,rm,.04 w5" TBK_LIBRARY=$V/a_w5.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
one "hap front,rm,.06 w5" TBK_LIBRARY=$V/a_w5.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.06 -- --lists haplotypes
one "hap whole,rm,.04 w4" TBK_LIBRARY=$V/b_w4.so TBK_FRONT=0 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
done
one "counters uniform" TBK_LIBRARY=$V/z_counters.so --
one "counters hap front,rm,.04" TBK_LIBRARY=$V/z_counters.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
one "counters hap front,rm,.08" TBK_LIBRARY=$V/z_counters.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
one "counters hap front,ms,.08" TBK_LIBRARY=$V/z_counters.so TBK_FRONT=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
} 2>&1 | tee gpurun_out/r3c_ab.log
exit 0
