#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
AB_ENVS="X=1" bash tools/gpu_ab.sh 2>&1 | grep haplo | tee gpurun_out/ab_unroll_hap.log
exit 0
