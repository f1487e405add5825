#!/bin/bash
# round 3, batch I: filter words over a list's back keys - parity, then same-box A/B on both list shapes
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu --maxfail=8 -q 2>&1 | tail -15 ) > gpurun_out/r3i_tests.log 2>&1
tail -12 gpurun_out/r3i_tests.log
export TBK_SKIP_BUILD=1
one() {  # label, env..., -- bench flags
  python - "$@" <<'PY'
import json, os, subprocess, sys
label = sys.argv[1]; rest = sys.argv[2:]; i = rest.index("--"); envs, flags = rest[:i], rest[i + 1:]
env = dict(os.environ); env.update(e.split("=", 1) for e in envs)
p = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-streaming", "--timed-path", "resident"] + flags, env=env, capture_output=True, text=True, timeout=900)
line = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
if not line:
    print(label, "FAILED", p.stderr[-600:]); sys.exit(0)
d = json.loads(line[-1]); c = d["config"]; r = d["roofline"]
print(label, "|", d["value"], "Gb/s  single ms", r["kernel_ms_avg"], "probe", r["whole_probe_ms_avg"], "| load", c["table_load"], c["bucket_select"], "|", c["line_layout"][:5], "| GB", round(c["table_bytes_per_gpu"] / 1e9, 1),
      "| builds", c["layout_builds"], "past", c["keys_past_their_half"], "behind", c["keys_behind_front"], "parity", d["parity"]["all_ranks_equal"], d["parity"]["count_checksum"][:2], flush=True)
PY
}
{
for round in 1 2; do
  for f in 1 0; do
    one "uniform ms .08 filter=$f" TBK_FILTER=$f --
    one "hap rm .04 filter=$f" TBK_FILTER=$f TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
    one "hap rm .08 filter=$f" TBK_FILTER=$f TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
    one "hap ms .08 filter=$f" TBK_FILTER=$f TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
    one "hap ms .04 filter=$f" TBK_FILTER=$f TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
  done
done
one "hap ms .06 filter=1" TBK_FILTER=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.06 -- --lists haplotypes
one "uniform ms .12 filter=1" TBK_FILTER=1 TBK_FRONT=1 TBK_TABLE_LOAD=0.12 --
one "uniform ms .16 filter=1" TBK_FILTER=1 TBK_FRONT=1 TBK_TABLE_LOAD=0.16 --
} 2>&1 | tee gpurun_out/r3i_ab.log
exit 0
