#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 900 python tools/measure_reader.py --reads 40000 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_unique.py -x -q --timeout 300 2>&1 | tail -2
exit 0
