#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
for cfg in "0 0.04 haplotypes" "0 0.03 haplotypes" "0 0.025 haplotypes" "0 0.04 haplotypes" "0 0.03 haplotypes" "0 0.03 uniform" "0 0.04 uniform"; do
  set -- $cfg
  echo -n "samp=$1 load=$2 $3: "
  TBK_MOD_SAMPLING=$1 TBK_TABLE_LOAD=$2 timeout 600 python bench.py --lists $3 --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['config']['table_bytes_per_gpu']/1e9, d['table_build_s'])"
done
exit 0
