#!/bin/bash
# same-box A/B of kernel variants (TBK_LIBRARY), interleaved rounds; extra bench flags via AB_FLAGS,
# environment per variant run via AB_ENVS (";"-separated, e.g. "TBK_MOD_SAMPLING=0;TBK_MOD_SAMPLING=1")
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
IFS=';' read -ra ENVS <<< "${AB_ENVS:-X=0}"
for round in 1 2; do
for lib in $(ls $V/*.so); do
  for e in "${ENVS[@]}"; do
  for lists in uniform haplotypes; do
    echo -n "$(basename $lib) $e $lists: "
    env $e TBK_LIBRARY=$lib timeout 600 python bench.py --lists $lists --steps 10 --warmup 2 --no-cpu-baseline --no-streaming $AB_FLAGS 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['value'], d['config']['table_load'], d['config']['bucket_select'], d['config'].get('keys_in_other_half_of_home_line'), d['config'].get('keys_outside_home_line'))"
  done
  done
done
done
exit 0
