#!/bin/bash
# round 3, batch F: the host-fed step after the per-batch memset went away; list parsing on the GPU; copy trace
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1800 python -m pytest tests/test_gpu_lists.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_cli.py -m gpu --maxfail=8 -q 2>&1 | tail -30 ) > gpurun_out/r3f_tests.log 2>&1
tail -25 gpurun_out/r3f_tests.log
export TBK_SKIP_BUILD=1
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "resident", (d.get("kernel_resident") or {}).get("gbases_per_s"), "probe ms", d["roofline"]["whole_probe_ms_avg"], "single ms", d["roofline"]["kernel_ms_avg"], "frac", d["roofline"]["frac"], "parity", d["parity"].get("all_ranks_equal"), d["parity"].get("gpu_equals_cpu"))
    print("   variants", json.dumps(d.get("pipeline_variants"))[:700])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
( time timeout 900 python bench.py ) > gpurun_out/r3f_bench_default.log 2>&1; grep '^{"metric"' gpurun_out/r3f_bench_default.log | tail -1 > gpurun_out/r3f_bench_default.json; show gpurun_out/r3f_bench_default.json
( timeout 900 python bench.py --reads-per-step 65536 --no-cpu-baseline --no-streaming ) > gpurun_out/r3f_bench_small.log 2>&1; grep '^{"metric"' gpurun_out/r3f_bench_small.log | tail -1 > gpurun_out/r3f_bench_small.json; show gpurun_out/r3f_bench_small.json
( timeout 900 python bench.py --gpus 2 --share-device --steps 10 --no-cpu-baseline ) > gpurun_out/r3f_bench_2ranks.log 2>&1; grep '^{"metric"' gpurun_out/r3f_bench_2ranks.log | tail -1 > gpurun_out/r3f_bench_2ranks.json; show gpurun_out/r3f_bench_2ranks.json
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3f_trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3f_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --min-timed-s 0 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/r3f_trace.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3f_trace -name "*stats*" | head; for f in $(find gpurun_out/r3f_trace -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv" | head -4); do echo == $f; head -8 $f | cut -c1-200; done
find gpurun_out/r3f_trace -name "*_trace.csv" -size +3M -delete
exit 0
