#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
run() { echo -n "$*: "; timeout 900 python bench.py "$@" --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['config']['bucket_select'], d['config']['table_load'], round(d['config']['table_bytes_per_gpu']/1e9,1), d['table_build_s'])"; }
run --kmers-per-list 100000000
run --k 31
run --k 32
run --k 31 --lists haplotypes
run --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 9830
run --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 9830 --lists haplotypes
run --read-len 150 --reads-per-step 6553600
run --read-len 150 --reads-per-step 6553600 --lists haplotypes
exit 0
