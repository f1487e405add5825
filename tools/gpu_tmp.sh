#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
run() { echo -n "$*: "; timeout 120 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --kmers-per-list 50000000 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_only_gbases_per_s'], d['bins'])"; }
run --read-len 500000000 --reads-per-step 2
run --read-len 500000000 --reads-per-step 2 --lists haplotypes
exit 0
