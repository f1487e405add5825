#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
run() { echo -n "W=$TBK_MINIMIZER_W M=$TBK_MINIMIZER_M $*: "; timeout 900 python bench.py "$@" --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['config']['bucket_select'], d['config']['table_load'])"; }
export TBK_MINIMIZER_W=8 TBK_MINIMIZER_M=17
run --k 32
export TBK_MINIMIZER_W=7 TBK_MINIMIZER_M=16
run --k 32
run --k 32 --lists haplotypes
export TBK_MINIMIZER_W=6 TBK_MINIMIZER_M=17
run --k 32 --lists haplotypes
export TBK_MINIMIZER_W=8 TBK_MINIMIZER_M=16
run --k 27
run --k 27 --lists haplotypes
run --k 23
export TBK_MINIMIZER_W=6 TBK_MINIMIZER_M=16
run --k 27
run --k 27 --lists haplotypes
exit 0
