#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for i in 1 2; do
( timeout 600 python tools/measure_cli.py --reads 200000 ) > gpurun_out/cli_plain_input.json 2> gpurun_out/cli_plain_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_plain_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']['total_s']) for m in ('gzip','plain')})"
done
( time python -m trio_binning_amd.classify_by_kmers --help > /dev/null ) 2>&1 | grep real
( timeout 600 python tools/measure_cli.py --reads 60000 --gz-input ) > gpurun_out/cli_gz_input.json 2> gpurun_out/cli_gz_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_gz_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']['total_s']) for m in ('gzip','plain')})"
timeout 900 python -m pytest tests/test_gpu_integration.py tests/test_gpu_unique.py -x -q -m gpu 2>&1 | tail -2
exit 0
