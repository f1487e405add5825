#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_lookahead2.log
exit 0
