#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 600 python tools/measure_cli.py --reads 200000 ) > gpurun_out/cli_plain_input.json 2> gpurun_out/cli_plain_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_plain_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
( timeout 600 python tools/measure_cli.py --reads 60000 --gz-input ) > gpurun_out/cli_gz_input.json 2> gpurun_out/cli_gz_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_gz_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
( timeout 600 python tools/measure_reader.py --qual hifi ) 2>/dev/null | tail -1
TBK_SCAN_TIMING=1 timeout 600 python tools/measure_reader.py --qual const 2>&1 | grep "tbk-scan\|plain" | tail -4 | cut -c1-400
exit 0
