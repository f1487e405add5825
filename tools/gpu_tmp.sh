#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python tools/measure_multi_cli.py | tee gpurun_out/multi_cli.json | cut -c1-900
exit 0
