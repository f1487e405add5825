#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 1500 python tools/measure_unique_cli.py 2>&1 | tail -2 | cut -c1-1200
rm -rf /tmp/tbk_unique_*
exit 0
