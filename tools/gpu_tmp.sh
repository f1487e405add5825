#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
for round in 1 2; do
for lib in $(ls $V/*.so); do
  for lists in uniform haplotypes; do
  echo -n "$(basename $lib) $lists: "
  TBK_LIBRARY=$lib timeout 600 python bench.py --lists $lists --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'])"
  done
done
done
exit 0
