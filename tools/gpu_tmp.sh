#!/bin/bash
# one-off: counting kernel with merged 64-bit adds
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_unique.py tests/test_gpu_scale.py -m gpu -k "counter or unique" --maxfail=4 -q 2>&1 | tail -6 ) 2>&1 | tail -8
export TBK_SKIP_BUILD=1
for r in 1 2; do
( timeout 600 python bench.py --path count --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel_ms_avg'], r['kernel_only_gbases_per_s'], r['window_starts_per_s'], r['atomic_adds_ceiling_Gps'], r['atomic_frac'])"
done
( timeout 600 python bench.py --path count ) 2>&1 | grep '^{"metric"' | tail -1 > gpurun_out/bench_count.json; python -c "
import json; d=json.load(open('gpurun_out/bench_count.json')); print(d['value'], d['parity'], d['cpu_baseline']['value'])"
exit 0
