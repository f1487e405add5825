#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
( time timeout 2400 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -4 ) 2>&1 | tail -8
echo "--- fuzz 1500 seeds"; TBK_FUZZ_SEEDS=1500 timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unique.py -m gpu -k fuzz --maxfail=3 -q 2>&1 | tail -2
timeout 900 python bench.py 2>&1 | grep '^{"metric"' | tail -1 > gpurun_out/bench_front.json; python -c "
import json; d=json.loads(open('gpurun_out/bench_front.json').read()); r=d['roofline']; print(d['value'], r['frac'], r['kernel_ms_avg'], d['config']['line_layout'], d['config']['keys_behind_front'], d['config']['layout_builds'], d['parity'], d['streaming']['prepacked']['gbases_per_s'], d['streaming']['packed_on_submit']['gbases_per_s'])"
timeout 900 python bench.py --lists haplotypes --no-cpu-baseline --no-streaming 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['frac'], r['kernel_ms_avg'], d['config']['line_layout'], d['config']['layout_builds'])"
exit 0
