#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 1 --share-device --kmers-per-list 50000000 > gpurun_out/two_ranks.log 2>&1
grep -v "^\[W" gpurun_out/two_ranks.log | head -60
exit 0
