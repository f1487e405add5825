#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
R=$GRAFT_REPO_ROOT
V=$R/trio_binning_amd/csrc/variants
cd /tmp
for lib in a_asc b_zigzag; do
  export TBK_LIBRARY=$V/$lib.so
  rm -rf $R/gpurun_out/pmcx_$lib
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmcx_$lib -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcx_$lib.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for lib in ("a_asc", "b_zigzag"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcx_{lib}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "probe_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(lib, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
rm -rf gpurun_out/pmcx_*
exit 0
