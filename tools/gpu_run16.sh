#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python bench.py ) > gpurun_out/bench_default.log 2>&1
grep '^{"metric"' gpurun_out/bench_default.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'])); print(json.dumps(d['cpu_baseline'])); print(d['parity'])"
tail -4 gpurun_out/bench_default.log | grep real
exit 0
