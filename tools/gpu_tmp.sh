#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
free -g | head -2
( time timeout 2400 python bench.py --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 9830 --steps 10 --warmup 2 --cpu-seconds 6 ) > gpurun_out/bench_c5.log 2>&1
tail -5 gpurun_out/bench_c5.log | cut -c1-1500
( time timeout 2400 python bench.py --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 9830 --steps 10 --warmup 2 --cpu-seconds 6 --lists haplotypes ) > gpurun_out/bench_c5_hap.log 2>&1
tail -5 gpurun_out/bench_c5_hap.log | cut -c1-1500
exit 0
