#!/bin/bash
# same-box A/B of kernel variants on both the bench workload and the realistic clustered lists
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
bash tools/gpu_ab.sh
for lib in $(ls $V/*.so); do
  echo "realistic $(basename $lib): $(TBK_LIBRARY=$lib timeout 900 python tools/measure_realistic.py 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: (v['gbases_per_s'], v.get('counters')) for k, v in d.items() if k.startswith('load')})")"
done
D=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/dbg/c_dbg.so
if [ -f $D ]; then
  TBK_LIBRARY=$D timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep tbk-counters
  echo "counters realistic: $(TBK_LIBRARY=$D timeout 900 python tools/measure_realistic.py 2>&1 | tail -1)"
fi
exit 0
