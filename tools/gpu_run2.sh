#!/bin/bash
# plumbing check of the N>1 launch on a 1-GPU box: two ranks under torch.distributed.run, both on
# device 0 (--share-device); the numbers are not a scaling result
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 5 --warmup 1 --share-device --no-cpu-baseline --kmers-per-list 100000000 2>&1 | tail -3
exit 0
