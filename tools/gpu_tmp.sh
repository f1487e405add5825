#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
export TBK_SKIP_BUILD=1
pr() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'])"; }
for r in 1 2 3; do
echo -n "minimizer: "; TBK_MOD_SAMPLING=0 timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | pr
echo -n "mod-sampling: "; TBK_MOD_SAMPLING=1 timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | pr
done
exit 0
