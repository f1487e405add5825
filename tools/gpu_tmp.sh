#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( TBK_PINFLATE_TIMING=1 timeout 600 python tools/measure_reader.py --qual hifi ) 2> gpurun_out/pinflate_timing.log | tail -1
grep "tbk-pinflate" gpurun_out/pinflate_timing.log | sed -n '3,6p'
( timeout 600 python tools/measure_reader.py --qual const ) 2>/dev/null | tail -1
( timeout 600 python tools/measure_cli.py --reads 60000 --gz-input ) > gpurun_out/cli_gz_input.json 2> gpurun_out/cli_gz_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_gz_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
( timeout 600 python tools/measure_cli.py --reads 200000 ) > gpurun_out/cli_plain_input.json 2> gpurun_out/cli_plain_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_plain_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
( timeout 900 python tools/measure_unique_cli.py --gzip ) 2>/dev/null | tail -1 | cut -c1-330
exit 0
