"""The N>1 plumbing of bench.py on CPU: two gloo ranks exercise the shard plan, the barrier
and the max/sum reductions the timing contract uses.  No GPU compute is involved (the data
path has no collective; ranks only exchange the timing scalars)."""
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT

import bench

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
d = bench.Dist(world)
d.barrier()
lo, hi = bench.shard_plan(1001, rank, world)
elapsed = 1.0 + rank          # rank 1 is the slow one
units = hi - lo
out = dict(rank=rank, lo=lo, hi=hi, tmax=d.reduce(elapsed, "MAX"), usum=d.reduce(units, "SUM"))
d.barrier()
d.close()
print(json.dumps(out))
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_plan_covers_everything_once():
    for total in (0, 1, 7, 1000, 65536):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_plan(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_gloo_ranks_reduce_timing(tmp_path):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e.decode()[-2000:]
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [o["tmax"] for o in outs] == [2.0, 2.0]          # max over ranks, seen by both
    assert [o["usum"] for o in outs] == [1001.0, 1001.0]    # whole-job units
    assert outs[0]["lo"] == 0 and outs[0]["hi"] == outs[1]["lo"] and outs[1]["hi"] == 1001


def test_single_rank_dist_is_a_noop():
    d = bench.Dist(1)
    d.barrier()
    assert d.reduce(3.5, "MAX") == 3.5 and d.reduce(7, "SUM") == 7
    d.close()
