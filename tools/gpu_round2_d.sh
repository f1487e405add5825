#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
AB_ENVS="TBK_MOD_SAMPLING=0;TBK_MOD_SAMPLING=1" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_nt_modsampling.log
exit 0
