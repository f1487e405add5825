#!/bin/bash
# round 3, batch E: the GPU suite on the pipeline / native loop / split kernels, the new bench line at N = 1 and 2 (shared device)
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu --maxfail=8 -q 2>&1 | tail -40 ) > gpurun_out/r3e_tests.log 2>&1
tail -25 gpurun_out/r3e_tests.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -4
export TBK_SKIP_BUILD=1
( time timeout 900 python bench.py ) > gpurun_out/r3e_bench_default.log 2>&1; grep '^{"metric"' gpurun_out/r3e_bench_default.log | tail -1 > gpurun_out/r3e_bench_default.json; tail -c 1500 gpurun_out/r3e_bench_default.log | head -c 600
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3e_bench_default.json"))
    print("DEFAULT value", d["value"], "resident", d.get("kernel_resident"), "\nroofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms_avg", "whole_probe_ms_avg", "share_of_the_batch_windows", "frac_P1_merged_table_reading")},
          "\nparity", d["parity"], "\ndevices", d["devices"], "\nvariants", d.get("pipeline_variants"), "\ncpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("default bench failed", e)
PY
( time timeout 900 python bench.py --gpus 2 --share-device --steps 10 ) > gpurun_out/r3e_bench_2ranks.log 2>&1; grep '^{"metric"' gpurun_out/r3e_bench_2ranks.log | tail -1 > gpurun_out/r3e_bench_2ranks.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3e_bench_2ranks.json"))
    print("2 RANKS value", d["value"], "resident", d.get("kernel_resident"), "parity", d["parity"], "devices", d["devices"], d.get("distinct_devices"), "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("2-rank bench failed", e); print(open("gpurun_out/r3e_bench_2ranks.log").read()[-1500:])
PY
( timeout 900 python bench.py --lists haplotypes --no-cpu-baseline --no-streaming ) > gpurun_out/r3e_bench_hap.log 2>&1; grep '^{"metric"' gpurun_out/r3e_bench_hap.log | tail -1 > gpurun_out/r3e_bench_hap.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3e_bench_hap.json"))
    print("HAP value", d["value"], "resident", d.get("kernel_resident"), d["config"]["bucket_select"], d["config"]["table_load"], d["config"]["line_layout"][:12], "parity", d["parity"])
except Exception as e:
    print("hap bench failed", e); print(open("gpurun_out/r3e_bench_hap.log").read()[-1500:])
PY
( timeout 600 python tools/measure_cli.py --reads 200000 ) > gpurun_out/r3e_cli_plain.json 2> gpurun_out/r3e_cli_plain.err; tail -c 1200 gpurun_out/r3e_cli_plain.json; tail -5 gpurun_out/r3e_cli_plain.err
exit 0
