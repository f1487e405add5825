#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
for v in "" variants/libtbk_u1w4.so variants/libtbk_u2w5.so variants/libtbk_u1w5.so variants/libtbk_u1w6.so variants/libtbk_u2w6.so ""; do
  if [ -n "$v" ]; then export TBK_LIBRARY=$GRAFT_REPO_ROOT/trio_binning_amd/csrc/$v; else unset TBK_LIBRARY; fi
  echo -n "${v:-default}: "
  timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['bins'])"
done
exit 0
