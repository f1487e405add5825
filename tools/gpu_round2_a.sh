#!/bin/bash
# round 2, first measurement pass: tests, the new bench legs, N>1 plumbing without torch
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu --maxfail=6 -q --durations=8 2>&1 | tail -30 ) > gpurun_out/gpu_tests.log 2>&1
tail -12 gpurun_out/gpu_tests.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -3 gpurun_out/bench_default.log
export TBK_SKIP_BUILD=1
( time timeout 600 python bench.py --gpus 2 --share-device --kmers-per-list 100000000 --reads-per-step 65536 --steps 5 --no-cpu-baseline ) > gpurun_out/bench_2ranks.log 2>&1
tail -3 gpurun_out/bench_2ranks.log
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-device --kmers-per-list 100000000 --reads-per-step 65536 --steps 5 --no-cpu-baseline ) > gpurun_out/bench_2ranks_torchrun.log 2>&1
tail -3 gpurun_out/bench_2ranks_torchrun.log
( time timeout 600 python bench.py --scaling strong --strong-reads 1000000 --steps 3 --no-cpu-baseline --no-streaming ) > gpurun_out/bench_strong.log 2>&1
tail -3 gpurun_out/bench_strong.log
( time timeout 600 python bench.py --path count --steps 4 --warmup 1 ) > gpurun_out/bench_count.log 2>&1
tail -3 gpurun_out/bench_count.log
exit 0
