#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu --maxfail=8 -q 2>&1 | tail -15 ) > gpurun_out/r3h_tests.log 2>&1
tail -12 gpurun_out/r3h_tests.log
bash tools/gpu_profile.sh
