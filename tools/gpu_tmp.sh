#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
export TBK_SKIP_BUILD=1
pr() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'], d['config']['table_load'], d['config']['table_bytes_per_gpu']/1e9, d.get('parity',{}).get('gpu_equals_cpu'))"; }
echo -n "C5-like: "; timeout 1200 python bench.py --k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 8192 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | pr
echo -n "k=32: "; timeout 600 python bench.py --k 32 --steps 15 --warmup 3 --cpu-seconds 2 2>&1 | tail -1 | pr
echo -n "C2: "; timeout 600 python bench.py --kmers-per-list 100000000 --steps 15 --warmup 3 --cpu-seconds 2 2>&1 | tail -1 | pr
timeout 900 python tools/measure_realistic.py 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('realistic', {k: (v['gbases_per_s'] if isinstance(v, dict) else v) for k, v in d.items()})"
exit 0
