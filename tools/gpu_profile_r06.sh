#!/bin/bash
# Round 6's reference profile (a trimmed tools/gpu_profile.sh: the probe kernels are round 5's): the default bench line, the rocprofv3
# kernel-trace stats of the same command, the FETCH_SIZE pass on both list shapes, configs[4]'s two tables with build timing (what a
# replica's pinned build costs), the counting path, ranks sharing the device, the end-to-end legs.
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; export TBK_SKIP_BUILD=1
R=$GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
C5="--k 31 --kmers-per-list 1000000000 --read-len 100000 --reads-per-step 39322 --steps 10 --warmup 2 --min-timed-s 3 --no-cpu-baseline --no-streaming --no-realistic --no-strong-leg"
( time TBK_BUILD_TIMING=1 timeout 1200 python bench.py $C5 ) > gpurun_out/bench_c5_uniform.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_c5_uniform.log | tail -1 > gpurun_out/bench_c5_uniform.json
( time TBK_BUILD_TIMING=1 timeout 1200 python bench.py $C5 --lists haplotypes ) > gpurun_out/bench_c5_haplotypes.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_c5_haplotypes.log | tail -1 > gpurun_out/bench_c5_haplotypes.json
grep "tbk build" gpurun_out/bench_c5_uniform.log gpurun_out/bench_c5_haplotypes.log > gpurun_out/build_timing_configs4.log
( time timeout 600 python bench.py --path count ) > gpurun_out/bench_count.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_count.log | tail -1 > gpurun_out/bench_count.json
cd /tmp
rm -rf $R/gpurun_out/pmc_* $R/gpurun_out/prof_trace*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --no-strong-leg > $R/gpurun_out/prof_trace.log 2>&1
FLAGS="--steps 4 --warmup 1 --min-timed-s 0 --no-cpu-baseline --no-streaming --no-realistic --no-strong-leg"
for lists in uniform haplotypes; do
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  tag=${lists}_$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --lists $lists $FLAGS > $R/gpurun_out/pmc_$tag.log 2>&1
done
done
cd $R
python tools/profile_summary.py gpurun_out > gpurun_out/profile_summary.log 2>&1; tail -c 1500 gpurun_out/profile_summary.log
find gpurun_out -name "*_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*counter_collection.csv" -size +2M -delete
( time timeout 900 python bench.py --gpus 8 --share-device --kmers-per-list 30000000 --reads-per-step 32768 --steps 10 --warmup 2 --min-timed-s 2 --no-streaming --cpu-seconds 2 --no-strong-leg ) > gpurun_out/bench_8ranks_shared_device.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_8ranks_shared_device.log | tail -1 > gpurun_out/bench_8ranks_shared_device.json
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-device --steps 10 --warmup 2 --min-timed-s 3 --no-strong-leg ) > gpurun_out/bench_2ranks_torchrun.log 2>&1; grep "^{\"metric\"" gpurun_out/bench_2ranks_torchrun.log | tail -1 > gpurun_out/bench_2ranks_torchrun.json
# end to end with lists and reads shaped like real trio-binning input (the entry layout, its default verification, gzip'ed bins by the GPU encoder)
( TBK_WRITE_TIMING=1 timeout 1500 python tools/measure_e2e.py --lists haplotypes --dir /dev/shm --out-dir /tmp --modes plain,gzip ) > gpurun_out/cli_configs1_haplotypes.json 2> gpurun_out/cli_configs1_haplotypes.err
rm -rf /dev/shm/tbk_e2e_* /tmp/tbk_e2e_*
# three rings on the one device: TBK_DEVICES=0,0,0 through the command line, gzip'ed bins
( TBK_WRITE_TIMING=1 timeout 1500 python tools/measure_e2e.py --dir /dev/shm --out-dir /tmp --modes gzip --devices 0,0,0 ) > gpurun_out/cli_configs1_3rings.json 2> gpurun_out/cli_configs1_3rings.err
rm -rf /dev/shm/tbk_e2e_* /tmp/tbk_e2e_*
python - <<'PY'
import json
for f in ("bench_default", "bench_c5_uniform", "bench_c5_haplotypes", "bench_count", "bench_8ranks_shared_device", "bench_2ranks_torchrun"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["value"], d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("parity") or {}).get("gpu_equals_cpu"), (d.get("strong_90gbp") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
cat gpurun_out/build_timing_configs4.log | cut -c1-200
exit 0
