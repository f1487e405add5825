#!/bin/bash
# guests in the other half (v1 layout): parity, then bench at several loads, guests on/off, both list shapes
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_multi.py -m gpu --maxfail=8 -q 2>&1 | tail -40 ) > gpurun_out/gpu_parity.log 2>&1
tail -8 gpurun_out/gpu_parity.log
AB_ENVS="TBK_TABLE_LOAD=0.04 TBK_GUESTS=0;TBK_TABLE_LOAD=0.04 TBK_GUESTS=1;TBK_TABLE_LOAD=0.08 TBK_GUESTS=0;TBK_TABLE_LOAD=0.08 TBK_GUESTS=1;TBK_TABLE_LOAD=0.12 TBK_GUESTS=1" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_guests.log
exit 0
