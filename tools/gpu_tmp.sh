#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( TBK_PINFLATE_TIMING=1 timeout 600 python tools/measure_reader.py --qual hifi ) 2> gpurun_out/pinflate_timing.log | tail -1
grep -c "tbk-pinflate" gpurun_out/pinflate_timing.log; grep "tbk-pinflate" gpurun_out/pinflate_timing.log | sed -n '2,6p'
( TBK_PINFLATE=0 timeout 600 python tools/measure_reader.py --qual hifi ) 2>/dev/null | tail -1
( timeout 600 python tools/measure_reader.py --qual const ) 2>/dev/null | tail -1
( timeout 600 python tools/measure_cli.py --reads 60000 --gz-input ) > gpurun_out/cli_gz_input.json 2> gpurun_out/cli_gz_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_gz_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
exit 0
