#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 600 python tools/measure_reader.py --qual hifi ) 2>/dev/null | tail -1
( TBK_FASTQ_SCAN=0 timeout 600 python tools/measure_reader.py --qual hifi ) 2>/dev/null | tail -1
( timeout 600 python tools/measure_reader.py --qual hifi --reads 4000000 --read-len 150 ) 2>/dev/null | tail -1
( TBK_FASTQ_SCAN=0 timeout 600 python tools/measure_reader.py --qual hifi --reads 4000000 --read-len 150 ) 2>/dev/null | tail -1
for args in "--gzip" "--gzip --split 8"; do
  ( timeout 900 python tools/measure_unique_cli.py $args ) 2>gpurun_out/unique.err | tail -1 | cut -c1-330
done
( timeout 600 python tools/measure_cli.py --reads 60000 --gz-input ) > gpurun_out/cli_gz_input.json 2> gpurun_out/cli_gz_input.err
python -c "
import json; d=json.load(open('gpurun_out/cli_gz_input.json')); print({m:(d[m]['wall_s'], d[m]['stages']) for m in ('gzip','plain')})"
exit 0
