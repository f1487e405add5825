#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 600 python tools/calib_cache.py 2>&1 | tail -14
exit 0
