#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 600 python -m pytest tests/test_gpu_scale.py -x -q --timeout 300 2>&1 | tail -8
exit 0
