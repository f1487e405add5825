#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/ab_merged_compare.log
echo "--- parity with merged"
TBK_SKIP_BUILD=1 TBK_LIBRARY=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants/merged.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
exit 0
