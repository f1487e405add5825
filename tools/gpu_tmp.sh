#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
export TBK_SKIP_BUILD=1
V=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/variants
pr() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['config']['bucket_select'])"; }
for r in 1 2 3; do
echo -n "HEAD minimizer: "; TBK_LIBRARY=$V/libtbk_a_head.so timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | pr
echo -n "new minimizer: "; TBK_MOD_SAMPLING=0 TBK_LIBRARY=$V/libtbk_b_new.so timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | pr
echo -n "new mod-sampling: "; TBK_MOD_SAMPLING=1 TBK_LIBRARY=$V/libtbk_b_new.so timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | pr
done
exit 0
