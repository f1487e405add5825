#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
for cfg in "6 0 0.6" "0 0 0.6" "6 16 0.6" "6 16 0.4" "6 16 0.3" "4 16 0.5" "0 0 0.4" "6 0 0.3"; do
  set -- $cfg
  echo -n "W=$1 M=$2 load=$3: "
  TBK_COUNT_W=$1 TBK_COUNT_M=$2 TBK_COUNT_LOAD=$3 timeout 600 python tools/measure_count.py --genome 200000000 --coverage 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['gbases_per_s'], d['table_GB'], d['table_load'])"
done
exit 0
