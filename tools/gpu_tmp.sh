#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
for cfg in "haplotypes 15000 65536" "haplotypes 16384 60000" "uniform 15000 65536" "haplotypes 15000 65536" "uniform 15000 65536" "haplotypes 1000 983040" "uniform 1000 983040"; do
  set -- $cfg
  echo -n "$1 L=$2: "
  timeout 600 python bench.py --lists $1 --read-len $2 --reads-per-step $3 --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['value'])"
done
exit 0
