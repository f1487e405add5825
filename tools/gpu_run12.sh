#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
export TBK_SKIP_BUILD=1
pr() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['config']['bucket_select'], d['config']['table_bytes_per_gpu']/1e9, d.get('parity'))"; }
echo -n "default: "; timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | pr
echo -n "C5-like k=31 2x1e9: "; timeout 1200 python bench.py --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 8192 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | pr
echo -n "C5-like W=8: "; TBK_MINIMIZER_W=8 timeout 1200 python bench.py --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 8192 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | pr
echo -n "k=32 2x3e8: "; timeout 600 python bench.py --k 32 --steps 8 --warmup 2 --cpu-seconds 3 2>&1 | tail -1 | pr
echo -n "k=31 2x3e8 W=8: "; TBK_MINIMIZER_W=8 timeout 600 python bench.py --k 31 --steps 8 --warmup 2 --cpu-seconds 3 2>&1 | tail -1 | pr
exit 0
