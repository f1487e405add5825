#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 600 python tools/measure_reader.py --qual hifi ) 2>/dev/null | tail -1 | tee gpurun_out/reader_hifi.json
( timeout 600 python tools/measure_cli.py --reads 200000 ) > gpurun_out/cli_plain_input.json 2> gpurun_out/cli_plain_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_plain_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
AB_ENVS="X=1" bash tools/gpu_ab.sh 2>&1 | grep uniform | tee gpurun_out/ab_samp_unroll.log
exit 0
