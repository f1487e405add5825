#!/bin/bash
# round 3, batch D: longer spans (m = 15, w = 7: fewer line switches), event counters on haplotype lists
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
print(label, "|", d["value"], "Gb/s  kernel_ms", r["kernel_ms_avg"], "| load", c["table_load"], c["bucket_select"], "|", c["line_layout"][:11], "| table GB", round(c["table_bytes_per_gpu"] / 1e9, 1),
      "| builds", c["layout_builds"], "past", c["keys_past_their_half"], "behind", c["keys_behind_front"], "| bins", d["bins"], flush=True)
for l in p.stderr.splitlines():
    if l.startswith("tbk-counters"): print("   ", l, flush=True)
PY
}
{
for round in 1 2; do
one "uniform m16w6 .08" --
one "uniform m15w7 .08" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=1 TBK_TABLE_LOAD=0.08 --
one "uniform m15w7 .04" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=1 TBK_TABLE_LOAD=0.04 --
one "uniform m15w7 .06" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=1 TBK_TABLE_LOAD=0.06 --
done
one "uniform m15w7 .08 whole" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=0 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08 --
one "uniform m15w7 rm .08" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 --
one "hap m15w7 front,rm,.04" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
one "hap m15w7 front,rm,.08" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
one "hap m15w7 front,ms,.04" TBK_MINIMIZER_M=15 TBK_MINIMIZER_W=7 TBK_FRONT=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
one "hap m16w6 front,rm,.04" TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
one "counters uniform" TBK_LIBRARY=$V/z_counters.so --
one "counters hap front,rm,.04" TBK_LIBRARY=$V/z_counters.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
one "counters hap front,rm,.08" TBK_LIBRARY=$V/z_counters.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
one "counters hap front,ms,.08" TBK_LIBRARY=$V/z_counters.so TBK_FRONT=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
} 2>&1 | tee gpurun_out/r3d_ab.log
exit 0
