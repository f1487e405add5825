#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
export TBK_SKIP_BUILD=1
for e in "X=1" "TBK_MINIMIZER_W=7 TBK_MINIMIZER_M=15 TBK_MOD_SAMPLING=1" "TBK_MINIMIZER_W=7 TBK_MINIMIZER_M=15 TBK_MOD_SAMPLING=0" "TBK_MINIMIZER_W=5 TBK_MINIMIZER_M=17 TBK_MOD_SAMPLING=1"; do
 for l in uniform haplotypes; do
  echo -n "$e $l: "
  env $e timeout 600 python bench.py --lists $l --steps 10 --warmup 2 --no-cpu-baseline --no-streaming 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'], d['config']['table_load'], d['config']['layout_builds'], d['config']['keys_past_their_half'])"
 done
done 2>&1 | tee gpurun_out/ab_w7.log
exit 0
