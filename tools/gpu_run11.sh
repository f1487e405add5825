#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
bash tools/gpu_ab.sh
