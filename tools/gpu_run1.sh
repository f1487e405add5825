#!/bin/bash
# First GPU contact: parity tests, smoke, a small bench, then the calibration sweep.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
( timeout 600 python bench.py --kmers-per-list 10000000 --reads-per-step 8192 --steps 5 --warmup 1 --cpu-seconds 3 2>&1 | tail -5 ) > gpurun_out/bench_small.log
( timeout 900 python bench.py --steps 10 --warmup 2 --calibrate --no-cpu-baseline 2>&1 | tail -5 ) > gpurun_out/bench_full_calib.log
tail -3 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench_small.log gpurun_out/bench_full_calib.log
