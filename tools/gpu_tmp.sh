#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
TBK_FUZZ_SEEDS=60 timeout 900 python -m pytest tests/test_gpu_unique.py -x -q --timeout 300 2>&1 | tail -3
for load in 0.6 0.4 0.25; do
echo -n "load $load: "
TBK_COUNT_LOAD=$load timeout 900 python tools/measure_count.py --genome 200000000 --coverage 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['gbases_per_s'], d['table_GB'], d['table_load'], d['parity'])"
done
exit 0
