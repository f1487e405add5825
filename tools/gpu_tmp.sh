#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 900 python tools/measure_reader.py --reads 40000 2>&1 | tail -1
timeout 900 python tools/measure_reader.py --reads 40000 --qual const 2>&1 | tail -1
exit 0
