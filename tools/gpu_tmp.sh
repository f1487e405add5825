#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
D=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/dbg/c_dbg.so
for cfg in "0 0.04" "1 0.04" "1 0.03" "0 0.03"; do
  set -- $cfg
  echo -n "samp=$1 load=$2 haplotypes: "
  TBK_MOD_SAMPLING=$1 TBK_TABLE_LOAD=$2 TBK_LIBRARY=$D timeout 600 python bench.py --lists haplotypes --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep tbk-counters
done
exit 0
