#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
exit 0
