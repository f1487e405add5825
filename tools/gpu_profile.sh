#!/bin/bash
# Reference profile of the round: default bench (with streaming legs, cpu baseline, parity), the
# haplotype-shaped lists, the counting path; rocprofv3 kernel trace stats; PMC passes (each counter
# set in its own run with --kernel-trace only).  tools/profile_summary.py turns gpurun_out/ into the
# files kept under profiles/rNN_*/.
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
R=$GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
( time timeout 900 python bench.py --lists haplotypes ) > gpurun_out/bench_haplotypes.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_haplotypes.log | tail -1 > gpurun_out/bench_haplotypes.json
( time timeout 600 python bench.py --path count ) > gpurun_out/bench_count.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_count.log | tail -1 > gpurun_out/bench_count.json
export TBK_SKIP_BUILD=1
( time timeout 900 python bench.py --scaling strong --strong-reads 3000000 --steps 5 --no-cpu-baseline --no-streaming ) > gpurun_out/bench_strong.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_strong.log | tail -1 > gpurun_out/bench_strong.json
FLAGS="--steps 4 --warmup 1 --min-timed-s 0 --no-cpu-baseline --no-streaming"
cd /tmp
rm -rf $R/gpurun_out/pmc_* $R/gpurun_out/prof_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --steps 10 --warmup 2 --min-timed-s 0 --no-cpu-baseline --no-streaming > $R/gpurun_out/prof_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace_count -- python3 $R/bench.py --path count --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_trace_count.log 2>&1
for lists in uniform haplotypes; do
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES"; do
  tag=${lists}_$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --lists $lists $FLAGS > $R/gpurun_out/pmc_$tag.log 2>&1
done
done
cd $R
python tools/profile_summary.py gpurun_out
for e in "TBK_MOD_SAMPLING=1" "TBK_MOD_SAMPLING=0" "TBK_TABLE_LOAD=0.04"; do for l in uniform haplotypes; do echo -n "$e $l: "; env $e timeout 600 python bench.py --lists $l --steps 10 --warmup 2 --no-cpu-baseline --no-streaming 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"roofline\"][\"kernel_ms_avg\"], d[\"config\"][\"bucket_select\"], d[\"config\"][\"table_load\"])"; done; done > gpurun_out/ab_final_rules.log 2>&1
# host side of the boundary
( timeout 600 python tools/measure_reader.py --qual hifi ) > gpurun_out/reader_hifi.json 2> gpurun_out/reader_hifi.err
( timeout 600 python tools/measure_cli.py --reads 200000 ) > gpurun_out/cli_plain_input.json 2> gpurun_out/cli_plain_input.err
tail -c 600 gpurun_out/reader_hifi.json; tail -c 900 gpurun_out/cli_plain_input.json
exit 0
