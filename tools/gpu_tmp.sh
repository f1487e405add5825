#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 900 python tools/measure_cli.py --reads 60000 --gz-input 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['gbases'], d['fastq_GB'], {m: (d[m]['wall_s'], d[m]['stages']['read_s'], d[m]['stages']['write_s'], d[m]['out_bytes']) for m in ('gzip','plain')})"
rm -rf /tmp/tbk_cli_*
exit 0
