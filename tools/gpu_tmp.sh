#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 900 python tools/measure_count.py --genome 200000000 --coverage 20 --dump /tmp/dump_kmers.txt 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['gbases_per_s'], d['dump'], d['parity'])"
exit 0
