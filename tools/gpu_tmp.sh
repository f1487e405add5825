#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" ) 2>&1 | tail -4
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-400
exit 0
