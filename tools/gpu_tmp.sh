#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
for round in 1 2; do
for mb in 0 4096 8192 16384 65536; do
  for lists in uniform haplotypes; do
  echo -n "max_blocks=$mb $lists: "
  TBK_PROBE_MAX_BLOCKS=$mb timeout 600 python bench.py --lists $lists --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'])"
  done
done
done
exit 0
