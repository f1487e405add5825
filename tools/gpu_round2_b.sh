#!/bin/bash
# v2 table layout: parity first, then same-box A/B of 4 vs 5 waves per SIMD, sampling rule, load
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_multi.py -m gpu --maxfail=8 -q 2>&1 | tail -40 ) > gpurun_out/gpu_parity.log 2>&1
tail -15 gpurun_out/gpu_parity.log
AB_ENVS="${AB_ENVS:-TBK_MOD_SAMPLING=0}" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_v2.log
exit 0
