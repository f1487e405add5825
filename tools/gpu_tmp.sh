#!/bin/bash
# scratch script for one-off gpurun experiments (edited per experiment; every step under `timeout`)
mkdir -p gpurun_out; export TMPDIR=/tmp TBK_SKIP_BUILD=1
R=$GRAFT_REPO_ROOT
timeout 600 python $R/tools/measure_count.py --genome 200000000 --coverage 20 --dump /tmp/dump_kmers.txt > $R/gpurun_out/count_kmers.log 2>&1
cd /tmp
rm -rf $R/gpurun_out/prof_count
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_count -- python3 $R/tools/measure_count.py --genome 200000000 --coverage 20 > $R/gpurun_out/prof_count.log 2>&1
cd $R
cp gpurun_out/prof_count/*/*_kernel_stats.csv gpurun_out/count_kernel_stats.csv
rm -rf gpurun_out/prof_count
tail -1 gpurun_out/count_kmers.log | cut -c1-200
exit 0
