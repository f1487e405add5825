"""The N>1 plumbing of bench.py on CPU: two ranks exercise the shard plan, the barrier and the
max/sum reductions the timing contract uses - through the file rendezvous bench.py uses by default
(no torch anywhere) and through a torch.distributed gloo group (TBK_BENCH_DIST=gloo).  No GPU compute
is involved (the data path has no collective; ranks only exchange the timing scalars)."""
import json
import os
import socket
import subprocess
import sys

import pytest

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
elapsed = 1.0 + rank          # the last rank is the slow one
units = hi - lo
out = dict(rank=rank, lo=lo, hi=hi, tmax=d.reduce(elapsed, "MAX"), usum=d.reduce(units, "SUM"), backend=d.backend,
           gathered=d.gather(10 * rank), objects=d.gather_obj(dict(rank=rank, device="0000:%02x:00.0" % rank, sum=[rank, 2 * rank])))
for i in range(50):           # many rounds back to back: nobody may overtake, no file may be read stale
    assert d.gather(i * world + rank) == [float(i * world + r) for r in range(world)]
# the region loop of bench.py: every rank takes the same number of regions although their own times differ
times = bench.timed_regions(lambda: 0.02 * (1 + rank), d, 0.11)
out["regions"] = len(times)
out["torch_loaded"] = "torch" in sys.modules
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


def _run_ranks(tmp_path, world, backend, rdv_env):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TBK_BENCH_DIST=backend, **rdv_env)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e.decode()[-2000:]
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    return outs


@pytest.mark.parametrize("backend,world", [("file", 2), ("file", 4), ("gloo", 2)])
def test_ranks_reduce_timing(tmp_path, backend, world):
    rdv = tmp_path / "rdv"
    # "file" with an explicit directory (bench.py's own spawner) ...
    outs = _run_ranks(tmp_path, world, backend, {"TBK_BENCH_RDV": str(rdv)} if backend == "file" else {})
    assert [o["tmax"] for o in outs] == [float(world)] * world           # max over ranks, seen by all
    assert [o["usum"] for o in outs] == [1001.0] * world                  # whole-job units
    assert all(o["gathered"] == [10.0 * r for r in range(world)] for o in outs)
    # what the bench line's per-rank device identities and parity checksums travel in
    assert all(o["objects"] == [{"rank": r, "device": "0000:%02x:00.0" % r, "sum": [r, 2 * r]} for r in range(world)] for o in outs)
    assert outs[0]["lo"] == 0 and outs[-1]["hi"] == 1001
    for a, b in zip(outs, outs[1:]):
        assert a["hi"] == b["lo"]
    assert len({o["regions"] for o in outs}) == 1 and outs[0]["regions"] >= 2
    assert all(o["backend"] == backend for o in outs)
    if backend == "file":
        assert not any(o["torch_loaded"] for o in outs)                   # the default path never imports torch
        assert not rdv.exists() or not os.listdir(rdv)                     # the rendezvous cleans up after itself


def test_file_rendezvous_under_a_launcher(tmp_path):
    """Without TBK_BENCH_RDV the directory is derived from what a launcher hands all its workers alike
    (MASTER_PORT and its own pid, the workers' parent) - how `python -m torch.distributed.run` starts bench.py."""
    env = {k: v for k, v in os.environ.items() if k != "TBK_BENCH_RDV"}
    outs = None
    import unittest.mock as m

    with m.patch.dict(os.environ, env, clear=True):
        outs = _run_ranks(tmp_path, 2, "file", {})
    assert [o["tmax"] for o in outs] == [2.0, 2.0] and [o["usum"] for o in outs] == [1001.0, 1001.0]


def test_single_rank_dist_is_a_noop():
    d = bench.Dist(1)
    d.barrier()
    assert d.reduce(3.5, "MAX") == 3.5 and d.reduce(7, "SUM") == 7 and d.gather(2) == [2.0]
    assert bench.timed_regions(lambda: 0.4, d, 1.0) == [0.4, 0.4, 0.4]
    d.close()


def test_metric_label_names_what_ran():
    assert bench.metric_label(21, 300_000_000) == "Gbases/sec classified (k=21, 2x300M k-mer tables)"   # BASELINE.json's
    assert bench.metric_label(31, 1_000_000_000) == "Gbases/sec classified (k=31, 2x1000M k-mer tables)"
    assert bench.metric_label(21, 100_000_000) == "Gbases/sec classified (k=21, 2x100M k-mer tables)"


def test_bench_source_never_imports_torch_by_default():
    src = open(os.path.join(ROOT, "bench.py")).read()
    lines = [l.strip() for l in src.splitlines() if "import torch" in l]
    # the only torch imports sit in the opt-in gloo backend of Dist
    assert lines == ["import torch.distributed as dist"]
    assert src.count("import torch") == 1
