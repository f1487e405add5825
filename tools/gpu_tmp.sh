#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 1500 python tools/measure_cli.py --reads 200000 2>&1 | tail -2
rm -rf /tmp/tbk_cli_*
exit 0
