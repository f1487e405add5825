#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 600 python tools/measure_reader.py 2>&1 | tail -1
TBK_INFLATE=zlib timeout 600 python tools/measure_reader.py 2>&1 | tail -1
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -2
exit 0
