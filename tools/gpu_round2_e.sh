#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_multi.py tests/test_gpu_scale.py -m gpu --maxfail=8 -q 2>&1 | tail -30 ) > gpurun_out/gpu_parity.log 2>&1
tail -12 gpurun_out/gpu_parity.log
export TBK_SKIP_BUILD=1
for lists in uniform haplotypes; do
  timeout 600 python bench.py --lists $lists --no-cpu-baseline --no-streaming 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lists', d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'], d['config']['layout_builds'], d['config']['keys_past_their_half'], d['table_build_s'])"
done
for cfg in "--k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 39320" "--kmers-per-list 100000000"; do
for lists in uniform haplotypes; do
  timeout 900 python bench.py $cfg --lists $lists --no-cpu-baseline --no-streaming 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg $lists', d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'], d['config']['layout_builds'], d['config']['keys_past_their_half'], d['table_build_s'], d['config']['table_load'])"
done
done
exit 0
