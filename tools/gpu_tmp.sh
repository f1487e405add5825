#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
for round in 1 2 3; do
for lib in $(ls $V/*.so); do
  echo -n "$(basename $lib): "
  TBK_LIBRARY=$lib timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
done
done
exit 0
