#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
export TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
for lib in $V/libtbk_a_inlinewalk.so $V/libtbk_b_queue.so; do
  echo "== $(basename $lib)"
  TBK_LIBRARY=$lib timeout 900 python tools/measure_realistic.py 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: (v['gbases_per_s'] if isinstance(v, dict) else v) for k, v in d.items()})"
  for r in 1 2; do
  TBK_LIBRARY=$lib timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['kernel_ms_avg'])"
  done
done
exit 0
