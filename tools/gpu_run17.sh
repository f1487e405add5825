#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python tools/measure_realistic.py 2>&1 | tail -1
exit 0
