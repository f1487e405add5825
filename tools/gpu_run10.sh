#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
export TBK_SKIP_BUILD=1
for cfg in "15000 65536" "16384 60000" "15000 65536"; do
  set -- $cfg
  echo -n "read_len=$1 reads=$2: "
  timeout 600 python bench.py --read-len $1 --reads-per-step $2 --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_only_gbases_per_s'], d['bins'])"
done
exit 0
