#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
pr() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'], d['config']['table_bytes_per_gpu']/1e9)"; }
for w in 6 8 6 8; do
echo -n "C5-like k=31 2x1e9 W=$w: "; TBK_MINIMIZER_W=$w timeout 1200 python bench.py --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 8192 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | pr
done
for w in 6 8; do
echo -n "k=31 2x3e8 W=$w: "; TBK_MINIMIZER_W=$w timeout 600 python bench.py --k 31 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | pr
done
exit 0
