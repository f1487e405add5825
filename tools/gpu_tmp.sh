#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_front_vs_head.log
exit 0
