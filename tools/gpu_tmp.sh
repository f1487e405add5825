#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for args in "" "--gzip" "--gzip --split 8"; do
  ( timeout 900 python tools/measure_unique_cli.py $args ) 2>gpurun_out/unique.err | tail -1
done
TBK_PINFLATE=0 timeout 900 python tools/measure_unique_cli.py --gzip 2>>gpurun_out/unique.err | tail -1
exit 0
