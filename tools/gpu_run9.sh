#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
echo "== C5-like: k=31, 2x1e9 keys, 100 kb reads"
timeout 1200 python bench.py --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 8192 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1800
echo "== C2: 2x100M keys"
timeout 600 python bench.py --kmers-per-list 100000000 --steps 10 --warmup 2 --cpu-seconds 4 2>&1 | tail -1 | cut -c1-2500
echo "== k=32"
timeout 600 python bench.py --k 32 --kmers-per-list 100000000 --steps 5 --warmup 1 --cpu-seconds 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('value',)} , d['config']['bucket_select'], d['parity'], d['roofline']['kernel_ms_avg'])"
exit 0
