#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
for round in 1 2; do
for samp in 0 1; do
  for lists in uniform haplotypes; do
  echo -n "samp=$samp $lists: "
  TBK_MOD_SAMPLING=$samp timeout 600 python bench.py --lists $lists --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'])"
  done
done
done
exit 0
