#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
for s in 1 8; do
timeout 1500 python tools/measure_unique_cli.py --gzip --split $s 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['files_per_parent'], d['file_GB_each_parent'], d['find_unique_s'], d['list_sizes'], d['binned_to_the_right_parent'])"
rm -rf /tmp/tbk_unique_*
done
exit 0
