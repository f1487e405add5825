#!/bin/bash
# round 3, batch A: parity of the front layout's back-half queue (both sampling rules), where a step's time
# goes (occupancy / no-load / cheap-bucket diagnostics), and the front layout on haplotype-shaped lists
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -8 ) > gpurun_out/r3a_tests.log 2>&1
for e in "TBK_FRONT=1 TBK_MOD_SAMPLING=0" "TBK_FRONT=1 TBK_MOD_SAMPLING=1" "TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_GUESTS=0"; do
  echo "== $e" >> gpurun_out/r3a_tests.log
  ( env $e timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu --maxfail=6 -q 2>&1 | tail -6 ) >> gpurun_out/r3a_tests.log 2>&1
done
tail -30 gpurun_out/r3a_tests.log
export TBK_SKIP_BUILD=1
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
PY
}
{
for round in 1 2; do
  for v in a_base d_occ3 d_occ2 d_noload d_cheap; do one "uniform $v" TBK_LIBRARY=$V/$v.so -- ; done
done
} 2>&1 | tee gpurun_out/r3a_diag.log
{
for round in 1 2; do
  one "hap old-default(whole,rm,.04)" TBK_FRONT=0 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
  one "hap front,rm,.04" TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
  one "hap front,rm,.08" TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
  one "hap front,ms,.08" TBK_FRONT=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
  one "hap front,ms,.04" TBK_FRONT=1 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
  one "hap whole,ms,.08" TBK_FRONT=0 TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
done
one "uniform front,ms,.04" TBK_TABLE_LOAD=0.04 --
one "uniform front,ms,.12" TBK_TABLE_LOAD=0.12 --
one "uniform front,ms,.16" TBK_TABLE_LOAD=0.16 --
one "uniform front,rm,.08" TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 --
} 2>&1 | tee gpurun_out/r3a_hap.log
exit 0
