#!/bin/bash
# v3b (lean fast path): parity, sweep, PMC instruction counters for W=6.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
export TBK_SKIP_BUILD=1
: > gpurun_out/sweep.log
for cfg in "0 0.25" "4 0.125" "6 0.125" "6 0.0625" "5 0.125"; do
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
R=$GRAFT_REPO_ROOT
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "probe" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[1], {k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
exit 0
