#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read()); r=d['roofline']; print(d['value'], r['frac'], r['kernel_ms_avg'], r['traffic'], r.get('traffic_GBps'), r.get('random_line_frac'), d['config']['line_layout'][:5], d['streaming']['prepacked']['gbases_per_s'], d['streaming']['packed_on_submit']['gbases_per_s'], d['streaming']['ascii']['gbases_per_s'], d['cpu_baseline']['value'], d['parity']['gpu_equals_cpu'])"
exit 0
