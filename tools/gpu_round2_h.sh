#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
AB_ENVS="TBK_TABLE_LOAD=0.04;TBK_TABLE_LOAD=0.0625;TBK_TABLE_LOAD=0.08;TBK_TABLE_LOAD=0.1" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_load_lean.log
exit 0
