#!/bin/bash
# one-off: packer with a persistent pool against threads per batch, interleaved on one box
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
export TBK_SKIP_BUILD=1
for r in 1 2 3; do for p in 1 0; do
echo -n "pool=$p: "
( TBK_PACK_POOL=$p timeout 600 python bench.py --no-cpu-baseline --steps 5 --min-timed-s 0 ) 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['streaming']; print(d['value'], 'stream', s['ascii']['gbases_per_s'], s['packed_on_submit']['gbases_per_s'], s['prepacked']['gbases_per_s'], s['prepacked']['host_pack_alone_gbases_per_s'])"
done; done
nproc; cat /sys/fs/cgroup/cpu.max; uptime
exit 0
