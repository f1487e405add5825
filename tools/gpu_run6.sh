#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
python tools/measure_streaming.py --pinned 1 2>&1 | tail -1
python tools/measure_streaming.py --pinned 0 2>&1 | tail -1
python tools/measure_streaming.py --pinned 1 --reads 65536 --batches 12 2>&1 | tail -1
exit 0
