#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
timeout 900 python tools/measure_cli.py 2>&1 | tail -2
exit 0
