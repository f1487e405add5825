#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 600 python tools/measure_reader.py 2>&1 | tail -1
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -2
timeout 900 python tools/measure_unique_cli.py --gzip --split 1 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['files_per_parent'], d['file_GB_each_parent'], d['find_unique_s'], d['list_sizes'], d['binned_to_the_right_parent'])"
rm -rf /tmp/tbk_unique_*
exit 0
