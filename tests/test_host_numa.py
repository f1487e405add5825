"""NUMA placement and the per-rank share of the host threads ("pinned, NUMA-local buffers, one
feeder thread per GPU").  CPU-only: the sysfs lookups read a fake tree under
TBK_SYSFS_ROOT, the binding is checked on a real thread's affinity mask."""
import ctypes as C
import os
import subprocess
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_sysfs(root, bdf_nodes, node_cpus):
    for bdf, node in bdf_nodes.items():
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True, exist_ok=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpulist in node_cpus.items():
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True, exist_ok=True)
        (d / "cpulist").write_text(cpulist + "\n")


def test_sysfs_lookups(built, tmp_path, monkeypatch):
    from trio_binning_amd._lib import lib

    _fake_sysfs(tmp_path, {"0000:c1:00.0": 1, "0000:05:00.0": 0, "0000:85:00.0": -1}, {0: "0-3,8,10-11", 1: "64-127,192-255"})
    monkeypatch.setenv("TBK_SYSFS_ROOT", str(tmp_path))
    assert lib.tbk_numa_node_of_pci_(b"0000:c1:00.0") == 1
    assert lib.tbk_numa_node_of_pci_(b"0000:C1:00.0") == 1       # hipDeviceGetPCIBusId spells hex digits either way
    assert lib.tbk_numa_node_of_pci_(b"0000:05:00.0") == 0
    assert lib.tbk_numa_node_of_pci_(b"0000:85:00.0") == -1      # the kernel says "unknown"
    assert lib.tbk_numa_node_of_pci_(b"0000:ff:00.0") == -1      # no such device
    assert lib.tbk_numa_node_of_pci_(b"") == -1
    cpus = (C.c_int * 512)()
    assert lib.tbk_numa_node_cpus_(0, cpus, 512) == 7 and list(cpus[:7]) == [0, 1, 2, 3, 8, 10, 11]
    assert lib.tbk_numa_node_cpus_(1, cpus, 512) == 128 and cpus[0] == 64 and cpus[63] == 127 and cpus[64] == 192 and cpus[127] == 255
    assert lib.tbk_numa_node_cpus_(1, cpus, 4) == 128             # the count does not depend on the capacity
    assert lib.tbk_numa_node_cpus_(7, cpus, 512) == 0 and lib.tbk_numa_node_cpus_(-1, cpus, 512) == 0


def _in_thread(fn):
    out = {}

    def run():
        out["v"] = fn()

    t = threading.Thread(target=run)
    t.start()
    t.join()
    return out["v"]


def test_thread_binding_and_fallbacks(built, tmp_path, monkeypatch):
    from trio_binning_amd._lib import lib

    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("one CPU: nothing to bind")
    half = allowed[: len(allowed) // 2]
    outside = max(allowed) + 1000
    _fake_sysfs(tmp_path, {}, {0: ",".join(map(str, half)), 1: ",".join(map(str, allowed)), 2: f"{outside}-{outside + 3}",
                                 3: ",".join(map(str, half + [outside]))})
    monkeypatch.setenv("TBK_SYSFS_ROOT", str(tmp_path))
    monkeypatch.delenv("TBK_NUMA", raising=False)
    before = os.sched_getaffinity(0)

    def bind(node):
        def go():
            n = lib.tbk_numa_bind_thread_(node)
            return n, sorted(os.sched_getaffinity(0))  # (pid 0: the calling thread)
        return _in_thread(go)

    assert bind(0) == (len(half), half)                      # bound to the node's CPUs
    assert bind(3) == (len(half), half)                      # ... to those of them the mask allows
    assert bind(1) == (len(allowed), allowed)                # already inside the node: unchanged
    assert bind(2) == (0, allowed)                           # nothing in common: left alone
    assert bind(9) == (0, allowed) and bind(-1) == (0, allowed)  # unknown node
    monkeypatch.setenv("TBK_NUMA", "0")
    assert bind(0) == (0, allowed)                           # switched off
    assert os.sched_getaffinity(0) == before                 # the caller's own thread was never touched


def test_pipeline_feeders_report_no_binding_without_devices(built):
    """Feeder threads bind themselves to their device's node; test rings have no device and stay where they are."""
    from trio_binning_amd._lib import lib

    SUBMIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64))
    WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64)
    sub = SUBMIT(lambda user, slot, bases, offs, n, counts, tk: 0)
    wt = WAIT(lambda user, slot, tk: 0)
    p = C.c_void_p()
    lib.tbk_pipeline_create_test_.argtypes = [C.c_int, C.c_int, SUBMIT, WAIT, C.c_void_p, C.POINTER(C.c_void_p)]
    assert lib.tbk_pipeline_create_test_(2, 2, sub, wt, None, C.byref(p)) == 0
    node, cpus = C.c_int(7), C.c_int(7)
    assert lib.tbk_pipeline_numa(p, 1, C.byref(node), C.byref(cpus)) == 0 and (node.value, cpus.value) == (-1, 0)
    assert lib.tbk_pipeline_numa(p, 2, C.byref(node), C.byref(cpus)) != 0
    lib.tbk_pipeline_destroy(p)


@pytest.mark.parametrize("env,expect", [
    ({}, lambda n: n),
    ({"LOCAL_WORLD_SIZE": "4"}, lambda n: max(1, n // 4)),     # torchrun: 4 ranks on this node share its CPUs
    ({"LOCAL_WORLD_SIZE": "64"}, lambda n: 1),
    ({"TBK_LOCAL_RANKS": "2", "LOCAL_WORLD_SIZE": "8"}, lambda n: max(1, n // 2)),  # the explicit knob wins over the launcher's
    ({"LOCAL_WORLD_SIZE": "4", "TBK_HOST_THREADS": "5"}, lambda n: 5),               # the override wins over everything
])
def test_host_threads_are_shared_between_local_ranks(built, env, expect):
    base = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "TBK_LOCAL_RANKS", "TBK_HOST_THREADS")}
    code = "import sys; sys.path.insert(0, %r); from trio_binning_amd._lib import lib; print(lib.tbk_host_threads())" % ROOT
    alone = int(subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, check=True).stdout.split()[-1])
    got = int(subprocess.run([sys.executable, "-c", code], env=dict(base, **env), capture_output=True, text=True, check=True).stdout.split()[-1])
    assert got == expect(alone), (env, alone, got)
