#!/bin/bash
# round 3, batch G: the host-fed step with a feeder that no longer sleeps on the oldest batch; end to end at configs[1] scale
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
export TBK_SKIP_BUILD=1
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "resident", (d.get("kernel_resident") or {}).get("gbases_per_s"), "probe ms", d["roofline"]["whole_probe_ms_avg"], "single ms", d["roofline"]["kernel_ms_avg"], "frac", d["roofline"]["frac"], "parity", d["parity"].get("all_ranks_equal"), d["parity"].get("gpu_equals_cpu"), d["config"]["bucket_select"], d["config"]["table_load"])
    print("   variants", json.dumps(d.get("pipeline_variants"))[:700])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
( time timeout 900 python bench.py ) > gpurun_out/r3g_bench_default.log 2>&1; grep '^{"metric"' gpurun_out/r3g_bench_default.log | tail -1 > gpurun_out/r3g_bench_default.json; show gpurun_out/r3g_bench_default.json
( timeout 900 python bench.py --gpus 2 --share-device --steps 10 --no-cpu-baseline ) > gpurun_out/r3g_bench_2ranks.log 2>&1; grep '^{"metric"' gpurun_out/r3g_bench_2ranks.log | tail -1 > gpurun_out/r3g_bench_2ranks.json; show gpurun_out/r3g_bench_2ranks.json
( timeout 900 python bench.py --lists haplotypes --no-cpu-baseline --no-streaming ) > gpurun_out/r3g_bench_hap.log 2>&1; grep '^{"metric"' gpurun_out/r3g_bench_hap.log | tail -1 > gpurun_out/r3g_bench_hap.json; show gpurun_out/r3g_bench_hap.json
( timeout 900 python bench.py --scaling strong --strong-reads 1500000 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming ) > gpurun_out/r3g_bench_strong.log 2>&1; grep '^{"metric"' gpurun_out/r3g_bench_strong.log | tail -1 > gpurun_out/r3g_bench_strong.json; show gpurun_out/r3g_bench_strong.json
df -h /tmp | tail -1; free -g | head -2
( time timeout 1500 python tools/measure_e2e.py ) > gpurun_out/r3g_e2e.json 2> gpurun_out/r3g_e2e.err; tail -c 3000 gpurun_out/r3g_e2e.json; tail -5 gpurun_out/r3g_e2e.err
exit 0
