#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
for load in 0.1 0.0625 0.05 0.04; do
  echo -n "bench load=$load: "
  TBK_TABLE_LOAD=$load timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['value'])"
done
TBK_LOADS=0.1,0.0625,0.05,0.04,0.03 timeout 600 python tools/measure_realistic.py 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: (v['gbases_per_s'], v['table_bytes']) for k, v in d.items() if k.startswith('load')})"
exit 0
