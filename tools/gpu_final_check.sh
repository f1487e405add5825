#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 2400 python -m pytest tests -m gpu --maxfail=6 -q --durations=6 2>&1 | tail -20 ) > gpurun_out/gpu_tests.log 2>&1
tail -12 gpurun_out/gpu_tests.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -4
( time timeout 900 python bench.py ) > gpurun_out/bench_final.log 2>&1
grep '^{"metric"' gpurun_out/bench_final.log | tail -1 | cut -c1-2500
( timeout 600 python tools/measure_reader.py --qual hifi ) 2>/dev/null | tail -1
( timeout 600 python tools/measure_cli.py --reads 60000 --gz-input ) > gpurun_out/cli_gz_input.json 2> gpurun_out/cli_gz_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_gz_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
( timeout 600 python tools/measure_cli.py --reads 200000 ) > gpurun_out/cli_plain_input.json 2> gpurun_out/cli_plain_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_plain_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
( timeout 900 python tools/measure_unique_cli.py --gzip ) 2>/dev/null | tail -1 | cut -c1-330
exit 0
