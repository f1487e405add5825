#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/calib_footprint.py 2>&1 | tail -9
exit 0
