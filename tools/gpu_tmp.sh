#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
TBK_FUZZ_SEEDS=2000 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unique.py -x -q -k fuzz --timeout 300 2>&1 | tail -3
exit 0
