#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_multi.py -m gpu --maxfail=8 -q 2>&1 | tail -12 ) 2>&1 | tail -8
AB_ENVS="X=1" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_lean_careful.log
AB_ENVS="TBK_TABLE_LOAD=0.08" bash tools/gpu_ab.sh 2>&1 | grep haplo | tee -a gpurun_out/ab_lean_careful.log
exit 0
