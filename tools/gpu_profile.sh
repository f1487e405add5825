#!/bin/bash
# Reference profile of the round: default bench (with cpu baseline), rocprofv3 kernel trace stats,
# PMC passes (FETCH_SIZE; TCC hit/miss; SQ instruction mix).  Summaries are copied to profiles/ by hand.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -4 gpurun_out/bench_default.log
( time timeout 900 python bench.py --lists haplotypes ) > gpurun_out/bench_haplotypes.log 2>&1
tail -4 gpurun_out/bench_haplotypes.log
export TBK_SKIP_BUILD=1
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pmc_* $R/gpurun_out/prof_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
cat gpurun_out/prof_trace/*/*_kernel_stats.csv | head -8
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("gpurun_out/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "probe_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["VGPR_Count"] = r["VGPR_Count"]; out["SGPR_Count"] = r["SGPR_Count"]; out["LDS_Block_Size"] = r["LDS_Block_Size"]
    for k, v in agg.items():
        out[k] = round(sum(v) / len(v))
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
print(out)
PY
exit 0
