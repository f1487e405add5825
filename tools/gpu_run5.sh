#!/bin/bash
# v3c: parity, sweep, PMC instruction counters for the default config.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
export TBK_SKIP_BUILD=1
: > gpurun_out/sweep.log
for cfg in "6 0.125"; do
  set -- $cfg
  echo "== W=$1 load=$2" >> gpurun_out/sweep.log
  TBK_MINIMIZER_W=$1 TBK_TABLE_LOAD=$2 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: print(line.strip()[:300]); continue
    print(json.dumps({k: d[k] for k in ('value','ms_per_step')} | {'kernel_ms': d['roofline']['kernel_ms_avg'], 'frac': d['roofline']['frac'], 'kernel_gb': d['roofline']['kernel_only_gbases_per_s'], 'sel': d['config']['bucket_select'], 'table_GB': d['config']['table_bytes_per_gpu']/1e9, 'build_s': d['table_build_s'], 'bins': d['bins']}))
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
exit 0
