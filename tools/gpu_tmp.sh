#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1
timeout 600 python bench.py --lists haplotypes > gpurun_out/bench_haplotypes.log 2>&1
exit 0
