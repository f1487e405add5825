#!/bin/bash
# Reference profile of the round: default bench (host-fed value, kernel_resident, pipeline variants, cpu baseline,
# parity), two ranks on the one device (plumbing of the N > 1 line), the haplotype-shaped lists, the counting
# path; rocprofv3 kernel trace stats of the same command as the default bench; PMC passes (each counter set in
# its own run with --kernel-trace only).  tools/profile_summary.py turns gpurun_out/ into the files kept under
# profiles/rNN/.
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; export TBK_SKIP_BUILD=1
R=$GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
export TBK_SKIP_BUILD=1
# the same lists in the key layout (what they got before short keys): the A/B on this box
( time TBK_SHORT=0 timeout 900 python bench.py --no-realistic --no-streaming --min-timed-s 5 --cpu-seconds 4 ) > gpurun_out/bench_default_key_layout.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_default_key_layout.log | tail -1 > gpurun_out/bench_default_key_layout.json
( time timeout 900 python bench.py --lists haplotypes ) > gpurun_out/bench_haplotypes.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_haplotypes.log | tail -1 > gpurun_out/bench_haplotypes.json
# BASELINE configs[4]'s table and read shape on one GPU: 2 x 1e9 31-mers (64-bit m-mer kernels), 100 kb reads: full keys (and the key layout they had), wide entries
C5="--k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 39322 --steps 10 --warmup 2 --min-timed-s 3 --no-cpu-baseline --no-streaming --no-realistic"
( time TBK_BUILD_TIMING=1 timeout 1200 python bench.py $C5 ) > gpurun_out/bench_c5_uniform.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_c5_uniform.log | tail -1 > gpurun_out/bench_c5_uniform.json
( time TBK_FULL=0 timeout 1200 python bench.py $C5 ) > gpurun_out/bench_c5_uniform_key_layout.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_c5_uniform_key_layout.log | tail -1 > gpurun_out/bench_c5_uniform_key_layout.json
( time TBK_BUILD_TIMING=1 timeout 1200 python bench.py $C5 --lists haplotypes ) > gpurun_out/bench_c5_haplotypes.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_c5_haplotypes.log | tail -1 > gpurun_out/bench_c5_haplotypes.json
# (same box: the full-key kernels' window loop unrolled 4 times - 66 registers, seven waves per SIMD - and not at all - 62, eight waves)
for v in full_u4 full_u1 full_u4 full_u1; do [ -f trio_binning_amd/csrc/variants/$v.so ] && TBK_LIBRARY=$R/trio_binning_amd/csrc/variants/$v.so timeout 600 python bench.py $C5 --no-sweep --steps 10 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['roofline']['kernel_ms_avg'], d['kernel_resident']['gbases_per_s'], d['value'])" ; done > gpurun_out/ab_full_unroll.log 2>&1
# configs[4] as SURVEY 8d writes it: log-normal read lengths, N50 100 kb, 150 Gbp, one rank
LN="--k 31 --kmers-per-list 1000000000 --read-lengths lognormal --read-len 66000 --scaling strong --strong-reads 2366000 --reads-per-step 60000 --steps 1 --warmup 0 --min-timed-s 0 --no-streaming --cpu-seconds 4 --parity-reads 512 --no-realistic"
( time timeout 2400 python bench.py $LN ) > gpurun_out/bench_c5_lognormal_uniform.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_c5_lognormal_uniform.log | tail -1 > gpurun_out/bench_c5_lognormal_uniform.json
( time timeout 2400 python bench.py $LN --lists haplotypes ) > gpurun_out/bench_c5_lognormal_haplotypes.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_c5_lognormal_haplotypes.log | tail -1 > gpurun_out/bench_c5_lognormal_haplotypes.json
( time timeout 600 python bench.py --path count ) > gpurun_out/bench_count.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_count.log | tail -1 > gpurun_out/bench_count.json
FLAGS="--steps 4 --warmup 1 --min-timed-s 0 --no-cpu-baseline --no-streaming --no-realistic"
cd /tmp
rm -rf $R/gpurun_out/pmc_* $R/gpurun_out/prof_*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py > $R/gpurun_out/prof_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace_count -- python3 $R/bench.py --path count --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_trace_count.log 2>&1
for lists in uniform haplotypes; do
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES"; do
  tag=${lists}_$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --lists $lists $FLAGS > $R/gpurun_out/pmc_$tag.log 2>&1
done
done
cd $R
python tools/profile_summary.py gpurun_out > gpurun_out/profile_summary.log 2>&1; tail -c 2500 gpurun_out/profile_summary.log
find gpurun_out -name "*_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*counter_collection.csv" -size +2M -delete
# the lines that REPLAY the traffic record (N > 1, several rings, strong scaling, the realistic_lists sub-record of the default line) run behind the
# PMC passes, on the record those passes have just made of these very kernels; the default line a second time for its sub-record
cp gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic_haplotypes.json profiles/ 2>/dev/null
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
# eight ranks on the one device (smaller lists: eight tables must fit its memory): the launch, the rendezvous, parity over all ranks, every rank's NUMA placement and share of the host threads
( time timeout 900 python bench.py --gpus 2 --share-device --min-timed-s 3 ) > gpurun_out/bench_2ranks_shared_device.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_2ranks_shared_device.log | tail -1 > gpurun_out/bench_2ranks_shared_device.json
( time timeout 900 python bench.py --gpus 8 --share-device --kmers-per-list 30000000 --reads-per-step 32768 --steps 10 --warmup 2 --min-timed-s 2 --no-streaming --cpu-seconds 2 ) > gpurun_out/bench_8ranks_shared_device.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_8ranks_shared_device.log | tail -1 > gpurun_out/bench_8ranks_shared_device.json
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-device --steps 10 --warmup 2 --min-timed-s 3 ) > gpurun_out/bench_2ranks_torchrun.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_2ranks_torchrun.log | tail -1 > gpurun_out/bench_2ranks_torchrun.json
( time timeout 900 python bench.py --rings 3 --no-cpu-baseline --no-streaming --no-realistic --min-timed-s 3 ) > gpurun_out/bench_3rings.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_3rings.log | tail -1 > gpurun_out/bench_3rings.json
# the literal BASELINE configs[2] line: the 90 Gbp set (6 M x 15 kb reads), one rank
( time timeout 1500 python bench.py --scaling strong --strong-reads 6000000 --steps 2 --warmup 1 --min-timed-s 0 --no-streaming --cpu-seconds 2 ) > gpurun_out/bench_strong.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_strong.log | tail -1 > gpurun_out/bench_strong.json
# host side of the boundary
( timeout 600 python tools/measure_reader.py --qual hifi ) > gpurun_out/reader_hifi.json 2> gpurun_out/reader_hifi.err
# end to end at configs[1] scale: inputs in memory (/dev/shm) and bins on the disk - the box's 84 GB file system then holds the 30 GB
# of bins only; and everything on that disk (34 GB of inputs + 5 GB of key caches + 30 GB of bins: ext4 runs low on free space
# against its dirty data and stops the writer for ~2 s near the end, profiles/r03/e2e_writer_stall.log)
( TBK_WRITE_TIMING=1 timeout 1500 python tools/measure_e2e.py --dir /dev/shm --out-dir /tmp --devices 0,0,0 ) > gpurun_out/cli_configs1.json 2> gpurun_out/cli_configs1.err
rm -rf /dev/shm/tbk_e2e_* /tmp/tbk_e2e_*
# the same with lists and reads shaped like real trio-binning input (the entry layout end to end)
( TBK_WRITE_TIMING=1 timeout 1500 python tools/measure_e2e.py --lists haplotypes --dir /dev/shm --out-dir /tmp --modes plain ) > gpurun_out/cli_configs1_haplotypes.json 2> gpurun_out/cli_configs1_haplotypes.err
rm -rf /dev/shm/tbk_e2e_* /tmp/tbk_e2e_*
# list loading at BASELINE configs[2] scale: 2 x 3e8-line text lists (13.2 GB), a small read set behind them
( timeout 1500 python tools/measure_e2e.py --kmers 300000000 --reads 100000 --modes plain --dir /dev/shm --out-dir /tmp ) > gpurun_out/cli_lists_configs2.json 2> gpurun_out/cli_lists_configs2.err
rm -rf /dev/shm/tbk_e2e_* /tmp/tbk_e2e_*
( TBK_WRITE_TIMING=1 timeout 1500 python tools/measure_e2e.py --modes plain ) > gpurun_out/cli_configs1_one_small_disk.json 2> gpurun_out/cli_configs1_one_small_disk.err
# .fastq.gz input at configs[1] scale: one ordinary gzip member and bgzf, HiFi-like qualities, bins plain
( timeout 2400 python tools/measure_e2e.py --dir /dev/shm --out-dir /tmp --modes plain --gz-input --qual hifi --gz-level 4 ) > gpurun_out/cli_gz_configs1.json 2> gpurun_out/cli_gz_configs1.err
rm -rf /dev/shm/tbk_e2e_* /tmp/tbk_e2e_*
( timeout 900 python tools/calib_ceilings.py ) > gpurun_out/calib_ceilings.json 2> gpurun_out/calib_ceilings.err
tail -c 600 gpurun_out/reader_hifi.json; tail -c 1500 gpurun_out/cli_configs1.json
exit 0
