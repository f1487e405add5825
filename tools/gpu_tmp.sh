#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
for round in 1 2 3; do
for cfg in "a_lean 0 haplotypes" "b_walk2 0 haplotypes" "a_lean 1 haplotypes" "b_walk2 1 haplotypes" "a_lean 0 uniform" "b_walk2 0 uniform"; do
  set -- $cfg
  echo -n "$1 samp=$2 $3: "
  TBK_MOD_SAMPLING=$2 TBK_LIBRARY=$V/$1.so timeout 600 python bench.py --lists $3 --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'])"
done
done
exit 0
