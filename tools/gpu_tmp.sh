#!/bin/bash
# one-off: the split-layout branch at 5 waves per SIMD with mod-sampling on uniform lists
export TMPDIR=/tmp TBK_SKIP_BUILD=1
cd v2wt
for e in "TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.08" "TBK_MOD_SAMPLING=0 TBK_TABLE_LOAD=0.08" "TBK_MOD_SAMPLING=1 TBK_TABLE_LOAD=0.04"; do
  echo -n "v2 5 waves $e uniform: "
  env $e timeout 600 python bench.py --lists uniform --steps 10 --warmup 2 --no-cpu-baseline --no-streaming 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'], d['config']['table_load'])"
done
exit 0
