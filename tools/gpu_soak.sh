#!/bin/bash
# soak: the seeded configuration fuzz with many seeds (every seed draws k, list shapes, read shapes,
# bucket-selection mode, table load, guests on/off; each runs with both PCIe transfers)
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time TBK_FUZZ_SEEDS=${SEEDS:-700} timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unique.py -m gpu -k "fuzz" --maxfail=5 -q 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/soak.log
exit 0
