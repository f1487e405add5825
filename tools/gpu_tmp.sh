#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
TBK_LIBRARY=$V/c_dbg.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep tbk-counters
TBK_LIBRARY=$V/c_dbg.so timeout 900 python tools/measure_realistic.py 2>&1 | tail -1
exit 0
