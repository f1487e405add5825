#!/bin/bash
# same-box A/B of kernel variants (TBK_LIBRARY), interleaved rounds
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
for round in 1 2; do
for lib in $(ls $V/*.so); do
  for cfg in "15000 65536" "16384 60000"; do
    set -- $cfg
    echo -n "$(basename $lib) L=$1: "
    TBK_LIBRARY=$lib timeout 600 python bench.py --read-len $1 --reads-per-step $2 --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'])"
  done
done
done
exit 0
