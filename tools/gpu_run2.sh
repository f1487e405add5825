#!/bin/bash
# v2 (paired 128-B bucket lines): parity, full bench with cpu baseline, rocprof kernel trace + PMC.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log
( timeout 900 python bench.py 2>&1 | tail -3 ) > gpurun_out/bench_full.log
export TBK_SKIP_BUILD=1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_trace.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_fetch.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_tcc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_tcc.log 2>&1 )
find gpurun_out -name "*.csv" | head -30
for f in gpurun_out/pytest_gpu.log gpurun_out/bench_full.log; do echo "== $f"; tail -3 $f; done
exit 0
