#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
AB_FLAGS="--lists uniform" bash tools/gpu_ab.sh 2>&1 | grep uniform | tee gpurun_out/ab_front_5waves.log
exit 0
