#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 2400 python -m pytest tests -m gpu --maxfail=6 -q --durations=15 2>&1 | tail -40 ) 2>&1 | tee gpurun_out/gpu_tests.log
exit 0
