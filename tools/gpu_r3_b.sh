#!/bin/bash
# round 3, batch B: the probe split into a single-read kernel (5 waves per SIMD) and a multi-read kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -8 ) > gpurun_out/r3b_tests.log 2>&1
tail -12 gpurun_out/r3b_tests.log
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
  for v in a_w5 b_w4 c_w5u2 d_w6; do one "uniform $v" TBK_LIBRARY=$V/$v.so -- ; done
done
for v in a_w5 b_w4; do
  one "hap whole,rm,.04 $v" TBK_LIBRARY=$V/$v.so TBK_FRONT=0 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
  one "hap front,rm,.04 $v" TBK_LIBRARY=$V/$v.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.04 -- --lists haplotypes
  one "hap front,rm,.08 $v" TBK_LIBRARY=$V/$v.so TBK_FRONT=1 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
  one "hap whole,rm,.08 $v" TBK_LIBRARY=$V/$v.so TBK_FRONT=0 TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08 -- --lists haplotypes
  one "uniform 150b reads $v" TBK_LIBRARY=$V/$v.so -- --read-len 150 --reads-per-step 8000000
  one "uniform 1kb reads $v" TBK_LIBRARY=$V/$v.so -- --read-len 1000 --reads-per-step 2000000
done
one "uniform a_w5 load .04" TBK_LIBRARY=$V/a_w5.so TBK_TABLE_LOAD=0.04 --
} 2>&1 | tee gpurun_out/r3b_ab.log
exit 0
