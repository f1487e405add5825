#!/usr/bin/env python3
"""End-to-end `classify-by-kmers` at BASELINE configs[1] scale: 1 M x 15 kb reads (15 Gbases, a 30 GB FASTQ)
against 2 x 100 M-line TEXT lists, from process start to the last byte of the three bins.

What is timed is the product's command line as a user runs it (a child process); the JSON says how long the
lists took (parse + table build, with and without the binary key cache), the stages' busy seconds inside the
native loop (reader / waiting for the GPU / writer) and the overall Gbases/s.  Inputs are synthetic (lists:
the bench's key generator; reads: its read generator, planted list k-mers) and are written by this script
just before the run, so the page cache is warm - stated in the output; nothing here can drop it.

    python tools/measure_e2e.py [--reads 1000000] [--kmers 100000000] [--gz-output] [--dir /tmp]
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--kmers", type=int, default=100_000_000)
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--read-len", type=int, default=15_000)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--dir", default=None)
ap.add_argument("--modes", default="plain,gzip")
ap.add_argument("--out-dir", default=None, help="where the runs write their bins (default: the inputs' temporary directory)")
ap.add_argument("--devices", default="", help="also run the plain mode with TBK_DEVICES set to this list (e.g. 0,0,0: three rings on one GPU)")
ap.add_argument("--keep", action="store_true")
ap.add_argument("--gz-input", action="store_true",
                help="also feed the reads as .fastq.gz, the way the reference's own CLI test and README do (tests/test_classify_by_kmers.py:19-36, seq.py:86-92): "
                     "once as ONE ordinary gzip member (a single DEFLATE chain: the reader's guessing inflater) and once as bgzf blocks, bins plain")
ap.add_argument("--gz-kinds", default="one_gzip_member,bgzf", help="which of the gzip'ed inputs to run")
ap.add_argument("--gz-trace", default="", help="a directory: the bgzf run once more under rocprofv3 --kernel-trace, its per-dispatch CSV kept there (when which kernels ran beside which)")
ap.add_argument("--gz-both", action="store_true",
                help="with --gz-input: every gzip'ed input also with gzip'ed bins - the reference's DEFAULT mode (seq.py:86-92 reads .gz through gzip.open, "
                     "classify_by_kmers.py:86-92 writes .gz unless --no-gzip-output): both ends compressed, sharing the host's CPUs")
ap.add_argument("--qual", choices=["const", "hifi"], default="const", help="quality strings: one symbol, or HiFi-like (60 %% at the cap, the rest spread: what a real .fastq.gz inflates like)")
ap.add_argument("--gz-level", type=int, default=6)
ap.add_argument("--gz-env", default="", help="further runs of the gzip'ed inputs under these environments, ';'-separated (e.g. TBK_PINFLATE_SPAN=4194304;TBK_PINFLATE_SPAN=8388608)")
ap.add_argument("--lists", choices=["uniform", "haplotypes"], default="uniform",
                help="haplotypes: lists and reads shaped like real trio-binning input (bench.py --lists haplotypes): the k-mers over the SNPs between two "
                     "haplotypes of an implicit genome, reads drawn from the haplotypes with errors")
a = ap.parse_args()

from trio_binning_amd import _lib, kmers  # noqa: E402
from trio_binning_amd._lib import check, lib  # noqa: E402

k, L, R, N = a.k, a.read_len, a.reads, a.kmers
tmp = tempfile.mkdtemp(prefix="tbk_e2e_", dir=a.dir)
out_root = tempfile.mkdtemp(prefix="tbk_e2e_out_", dir=a.out_dir) if a.out_dir else tmp
def fs_of(path):
    st = os.statvfs(path)
    return {"path": path, "total_GB": round(st.f_blocks * st.f_frsize / 1e9, 1), "free_GB_at_start": round(st.f_bavail * st.f_frsize / 1e9, 1)}


res = {"config": f"BASELINE configs[1] shape: {R} x {L} b reads, 2 x {N} unique {k}-mers as text lists", "dir": tmp,
       "inputs_on": fs_of(tmp), "outputs_on": fs_of(out_root),
       "page_cache": "warm", "host_usable_cpus": int(lib.tbk_host_threads())}
dev = 0


def dalloc(n):
    p = C.c_void_p()
    check(lib.tbk_device_alloc(dev, n, C.byref(p)))
    return p.value


# ---- the two lists as text ----------------------------------------------------------------------
t0 = time.time()
hap = a.lists == "haplotypes"
if hap:
    snp_rate, err_rate = 1 / 500, 0.002
    snp24, err24 = int(round(snp_rate * (1 << 24))), int(round(err_rate * (1 << 24)))
    genome_len = int(N / (1 - (1 - (2 * snp_rate - snp_rate ** 2 * (1 + 1 / 3))) ** k))
    cap = int(N * 1.05) + 1024
    d_keys = dalloc(2 * cap * 8)
    n_got = C.c_uint64()
    check(lib.tbk_synth_hap_keys_device(dev, 0x5EED0001, genome_len, snp24, k, C.c_void_p(d_keys), C.c_void_p(d_keys + cap * 8), cap, C.byref(n_got)))
    assert n_got.value <= cap
    N, key_stride = n_got.value, cap
    res["list_shape"] = f"haplotype-shaped: 2 x {N} k-mers over the SNPs (rate {snp_rate:g}) between two haplotypes of a {genome_len}-base genome; reads from the haplotypes, error rate {err_rate:g}"
else:
    d_keys = dalloc(2 * N * 8)
    check(lib.tbk_synth_keys_device(dev, 0x5EED0001, 0, 2 * N, k, C.c_void_p(d_keys)))
    key_stride = N
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
paths = []
for which in range(2):
    path = os.path.join(tmp, "hap%s.txt" % "AB"[which])
    paths.append(path)
    with open(path, "wb") as fh:
        step = 10_000_000
        for lo in range(0, N, step):
            n = min(step, N - lo)
            keys = np.empty(n, dtype=np.uint64)
            check(lib.tbk_memcpy_d2h(dev, keys.ctypes.data, C.c_void_p(d_keys + (which * key_stride + lo) * 8), n * 8))
            out = np.empty((n, k + 1), dtype=np.uint8)
            for i in range(k):
                out[:, i] = lut[((keys >> np.uint64(2 * i)) & np.uint64(3)).astype(np.intp)]
            out[:, k] = 10
            out.tofile(fh)
check(lib.tbk_device_free(dev, C.c_void_p(d_keys)))
res["lists_GB"] = round(sum(os.path.getsize(p) for p in paths) / 1e9, 2)

# ---- the reads as FASTQ ----------------------------------------------------------------------------
fq = os.path.join(tmp, "reads.fastq")
chunk = 20_000
d_b, d_o = dalloc(chunk * L + 64), dalloc((chunk + 1) * 8)
qual = np.full(L, ord("I"), dtype=np.uint8)
qual_pool = None
if a.qual == "hifi":
    # a pool of 2048 HiFi-like strings, one drawn per read: repeats lie megabytes apart, far outside DEFLATE's 32 KiB window
    qrng = np.random.default_rng(7)
    qv = np.clip(qrng.normal(60, 15, (2048, L)), 2, 93).astype(np.uint8)
    qv[qrng.random((2048, L)) < 0.6] = 93
    qual_pool = qv + 33
with open(fq, "wb") as fh:
    for first in range(0, R, chunk):
        n = min(chunk, R - first)
        if hap:
            check(lib.tbk_synth_hap_reads_device(dev, 0x5EED0001, genome_len, snp24, 0x5EED0002, first, n, L, err24, C.c_void_p(d_b), C.c_void_p(d_o)))
        else:
            check(lib.tbk_synth_reads_device(dev, 0x5EED0002, first, n, L, 0x5EED0001, N, N, k, 30, 3, C.c_void_p(d_b), C.c_void_p(d_o)))
        bases = np.empty((n, L), dtype=np.uint8)
        check(lib.tbk_memcpy_d2h(dev, bases.ctypes.data, C.c_void_p(d_b), n * L))
        names = [b"@read%09d c\n" % (first + i) for i in range(n)]  # fixed-width names: one record layout for the whole chunk
        w = len(names[0])
        rec = np.empty((n, w + L + 3 + L + 1), dtype=np.uint8)
        rec[:, :w] = np.frombuffer(b"".join(names), dtype=np.uint8).reshape(n, w)
        rec[:, w:w + L] = bases
        rec[:, w + L:w + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, w + L + 3:w + 2 * L + 3] = qual if qual_pool is None else qual_pool[(np.arange(first, first + n) * 2654435761 >> 7) % 2048]
        rec[:, -1] = 10
        rec.tofile(fh)
check(lib.tbk_device_free(dev, C.c_void_p(d_b)))
check(lib.tbk_device_free(dev, C.c_void_p(d_o)))
res["fastq_GB"] = round(os.path.getsize(fq) / 1e9, 2)
res["gbases"] = R * L / 1e9
res["inputs_written_s"] = round(time.time() - t0, 1)
t_sync = time.time()
os.sync()  # the inputs are at rest before anything is timed: their dirty pages count against the container's dirty limit, and a run that starts beside 34 GB of them has its writer throttled to the disk's speed
res["inputs_synced_s"] = round(time.time() - t_sync, 1)
res["page_cache"] = "warm (the inputs were written by this script just before the runs, then sync'ed: clean pages)"

# ---- the same reads as .fastq.gz ----------------------------------------------------------------------
gz_inputs = {}
if a.gz_input:
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    n_thr = max(2, int(lib.tbk_host_threads()))
    t_gz = time.time()

    def deflate_piece(args):
        piece, last, level = args
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return c.compress(piece) + c.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(piece)

    def bgzf_blocks(piece):
        out = bytearray()
        for i in range(0, len(piece), 60000):
            blk = piece[i:i + 60000]
            c = zlib.compressobj(a.gz_level, zlib.DEFLATED, -15)
            body = c.compress(blk) + c.flush()
            out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk))
        return bytes(out)

    one, bg = os.path.join(tmp, "reads_one_member.fastq.gz"), os.path.join(tmp, "reads_bgzf.fastq.gz")
    size = os.path.getsize(fq)
    piece_len = 8 << 20
    crc = 0
    # ONE gzip member whose DEFLATE chain was compressed piece by piece (pigz's way: every piece ends in an empty stored block, only the
    # last one is final): one chain to inflate, matches reach back across block borders inside a piece, nothing marks the blocks
    with open(fq, "rb") as src, open(one, "wb") as f1, open(bg, "wb") as f2, ThreadPoolExecutor(n_thr) as pool:
        f1.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff")
        done = 0
        while done < size:
            pieces = []
            for _ in range(n_thr * 2):
                b = src.read(piece_len)
                if not b:
                    break
                pieces.append(b)
            last_at = done + sum(len(b) for b in pieces) >= size
            outs = list(pool.map(deflate_piece, [(b, last_at and i == len(pieces) - 1, a.gz_level) for i, b in enumerate(pieces)]))
            blocks = list(pool.map(bgzf_blocks, pieces))
            for (body, _), b in zip(outs, pieces):
                f1.write(body)
                crc = zlib.crc32(b, crc)
            for blk in blocks:
                f2.write(blk)
            done += sum(len(b) for b in pieces)
        f1.write(struct.pack("<II", crc & 0xFFFFFFFF, size & 0xFFFFFFFF))
        f2.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # bgzf's empty end block
    gz_inputs = {"one_gzip_member": one, "bgzf": bg}
    res["gz_inputs"] = {"level": a.gz_level, "qualities": a.qual, "one_gzip_member_GB": round(os.path.getsize(one) / 1e9, 2), "bgzf_GB": round(os.path.getsize(bg) / 1e9, 2),
                        "written_s": round(time.time() - t_gz, 1),
                        "note": "one_gzip_member: a single gzip member, its DEFLATE chain compressed in 8 MB pieces by " + str(n_thr) + " threads (pigz's way): the reader sees one ordinary "
                                "stream without block markers; bgzf: 60 kB blocks, each a gzip member with a BC field"}
    os.sync()

# ---- list loading alone, three ways (in this process) ----------------------------------------------
for label, env in (("gpu_parser", {"TBK_LIST_CACHE": "0"}), ("host_parser", {"TBK_LIST_CACHE": "0", "TBK_LIST_GPU_PARSE": "0"}),
                   ("gpu_parser_writing_the_cache", {"TBK_LIST_CACHE": "1"}), ("key_cache", {})):
    for key in ("TBK_LIST_CACHE", "TBK_LIST_GPU_PARSE"):
        os.environ.pop(key, None)
    os.environ.update(env)
    t = time.perf_counter()
    ha, hb = kmers.HashSet.from_file(paths[0], dev), kmers.HashSet.from_file(paths[1], dev)
    t_lists = time.perf_counter() - t
    origin = ha.origin
    t = time.perf_counter()
    with kmers.Classifier(ha, hb):
        check(lib.tbk_device_sync(dev))
    t_table = time.perf_counter() - t
    res.setdefault("lists", {})[label] = {"origin": origin, "both_lists_s": round(t_lists, 3), "Mlines_per_s": round(2 * N / t_lists / 1e6, 1),
                                           "paired_table_build_s": round(t_table, 3)}
    ha.close(); hb.close()
for key in ("TBK_LIST_CACHE", "TBK_LIST_GPU_PARSE"):
    os.environ.pop(key, None)

# ---- the command line ---------------------------------------------------------------------------------
env = dict(os.environ, PYTHONPATH=ROOT, TBK_STATS="1", TBK_WRITE_TIMING="1")
runs = [(mode, cache) for mode in a.modes.split(",") for cache in ("text_lists", "cached_lists")]
if a.devices:
    runs.append(("plain", "devices_" + a.devices.replace(",", "_")))
for mode, cache in runs:
    if True:
        e = dict(env, TBK_LIST_CACHE="0") if cache == "text_lists" else dict(env)
        if cache.startswith("devices_"):
            e["TBK_DEVICES"] = a.devices
        out = os.path.join(out_root, mode + "_" + cache)
        os.makedirs(out)
        tsv = os.path.join(out, "stdout.tsv")
        t = time.time()
        with open(tsv, "wb") as so:
            p = subprocess.run([sys.executable, "-m", "trio_binning_amd.classify_by_kmers", fq, paths[0], paths[1],
                                "--haplotype-a-out-prefix", os.path.join(out, "hapA"), "--haplotype-b-out-prefix", os.path.join(out, "hapB"),
                                "--unclassified-out-prefix", os.path.join(out, "unc")] + (["--no-gzip-output"] if mode == "plain" else []),
                               env=e, stdout=so, stderr=subprocess.PIPE)
        dt = time.time() - t
        if p.returncode != 0:
            res[mode + "_" + cache] = {"failed": p.stderr.decode()[-1500:]}
            continue
        st = [l for l in p.stderr.decode().splitlines() if l.startswith("tbk-stats ")]
        stages = json.loads(st[-1][10:]) if st else {}
        wt = [l for l in p.stderr.decode().splitlines() if l.startswith("tbk-write-timing ")]
        lt = [l for l in p.stderr.decode().splitlines() if l.startswith("tbk-loop-timing ")]
        bins = {}
        with open(tsv, "rb") as fh:
            for line in fh:
                b = line.split(b"\t")[1].decode()
                bins[b] = bins.get(b, 0) + 1
        res[mode + "_" + cache] = {
            "wall_s": round(dt, 2), "gbases_per_s_wall": round(R * L / 1e9 / dt, 3), "stages": stages,
            "classify_loop_gbases_per_s": round(R * L / 1e9 / stages["loop_s"], 3) if stages.get("loop_s") else None,
            "before_the_loop_s": round(dt - stages.get("loop_s", 0), 2), "bins": bins, "write_timing": wt[-1] if wt else None, "loop_timing": lt[-1] if lt else None,
            "gpu_gzip": next((l for l in p.stderr.decode().splitlines() if l.startswith("tbk-gpu-gzip ")), None),
            "out_GB": round(sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) / 1e9, 2)}
        shutil.rmtree(out, ignore_errors=True)
        os.sync()  # the next run does not inherit this one's dirty pages
# ---- gzip'ed reads in, plain bins out ---------------------------------------------------------------------
if a.gz_trace and "bgzf" in gz_inputs:
    out = os.path.join(out_root, "gz_trace")
    os.makedirs(out)
    os.makedirs(a.gz_trace, exist_ok=True)
    with open(os.path.join(out, "stdout.tsv"), "wb") as so:
        p = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", a.gz_trace, "-o", "bgzf", "--", sys.executable, "-m", "trio_binning_amd.classify_by_kmers",
                            gz_inputs["bgzf"], paths[0], paths[1], "--haplotype-a-out-prefix", os.path.join(out, "hapA"), "--haplotype-b-out-prefix", os.path.join(out, "hapB"),
                            "--unclassified-out-prefix", os.path.join(out, "unc"), "--no-gzip-output"], env=dict(env, TMPDIR="/tmp"), stdout=so, stderr=subprocess.PIPE, cwd="/tmp")
    res["gz_trace"] = {"rc": p.returncode, "stderr_tail": p.stderr.decode()[-600:]}
    shutil.rmtree(out, ignore_errors=True)
gz_inputs = {label: path for label, path in gz_inputs.items() if label in a.gz_kinds.split(",")}
gz_runs = [(label, path, "", False) for label, path in gz_inputs.items()]
gz_runs += [(label, path, e, False) for e in a.gz_env.split(";") if e for label, path in gz_inputs.items()]
if a.gz_both:
    gz_runs += [(label, path, "", True) for label, path in gz_inputs.items()]
    gz_runs += [(label, path, e, True) for e in a.gz_env.split(";") if e for label, path in gz_inputs.items()]
for label, path, extra, gz_out in gz_runs:
    label = label + ("_gz_bins" if gz_out else "") + ("" if not extra else "_" + extra.replace("=", "_"))
    out = os.path.join(out_root, "gz_" + label)
    os.makedirs(out)
    tsv = os.path.join(out, "stdout.tsv")
    t = time.time()
    with open(tsv, "wb") as so:
        p = subprocess.run([sys.executable, "-m", "trio_binning_amd.classify_by_kmers", path, paths[0], paths[1],
                            "--haplotype-a-out-prefix", os.path.join(out, "hapA"), "--haplotype-b-out-prefix", os.path.join(out, "hapB"),
                            "--unclassified-out-prefix", os.path.join(out, "unc")] + ([] if gz_out else ["--no-gzip-output"]),
                           env=dict(env, TBK_PINFLATE_TIMING="1", **dict(kv.split("=", 1) for kv in extra.split(",") if kv)), stdout=so, stderr=subprocess.PIPE)
    dt = time.time() - t
    err = p.stderr.decode()
    if p.returncode != 0:
        res["gz_" + label] = {"failed": err[-1500:]}
        continue
    st = [l for l in err.splitlines() if l.startswith("tbk-stats ")]
    stages = json.loads(st[-1][10:]) if st else {}
    rounds = [l for l in err.splitlines() if l.startswith("tbk-pinflate round")]
    inflate = None
    if rounds:
        import re
        g = d = r = mb = 0.0
        guesses = kept = 0
        for l in rounds:
            m = re.search(r"(\d+) guesses, (\d+) chunks kept, ([\d.]+) MB of text: guess ([\d.]+) ms, decode ([\d.]+) ms, resolve ([\d.]+) ms", l)
            if m:
                guesses += int(m.group(1)); kept += int(m.group(2)); mb += float(m.group(3)); g += float(m.group(4)); d += float(m.group(5)); r += float(m.group(6))
        thr = int(lib.tbk_host_threads())
        inflate = {"rounds": len(rounds), "guesses": guesses, "chunks_kept": kept, "text_GB": round(mb / 1e3, 2), "guess_s": round(g / 1e3, 3), "decode_s": round(d / 1e3, 3), "resolve_s": round(r / 1e3, 3),
                   "threads": thr, "decode_MB_per_s_per_thread": round(mb / max(d / 1e3, 1e-9) / thr, 1), "resolve_MB_per_s_per_thread": round(mb / max(r / 1e3, 1e-9) / thr, 1),
                   "inflater_wall_GB_per_s": round(mb / 1e3 / max((g + d + r) / 1e3, 1e-9), 2)}
    bins = {}
    with open(tsv, "rb") as fh:
        for line in fh:
            b = line.split(b"\t")[1].decode()
            bins[b] = bins.get(b, 0) + 1
    res["gz_" + label] = {"wall_s": round(dt, 2), "gbases_per_s_wall": round(R * L / 1e9 / dt, 3), "stages": stages,
                          "classify_loop_gbases_per_s": round(R * L / 1e9 / stages["loop_s"], 3) if stages.get("loop_s") else None, "inflate": inflate, "bins": bins,
                          "text_GB_per_s_in_the_loop": round(res["fastq_GB"] / stages["loop_s"], 2) if stages.get("loop_s") else None,
                          "tbk_lines": [l for l in err.splitlines() if l.startswith(("tbk-gpu", "tbk-write", "tbk-loop", "tbk-read"))],
                          "bins_written_as": "gzip members" if gz_out else "plain text", "gpu_gzip": next((l for l in err.splitlines() if l.startswith("tbk-gpu-gzip ")), None),
                          "out_GB": round(sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) / 1e9, 2)}
    shutil.rmtree(out, ignore_errors=True)
    os.sync()
if not a.keep:
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(out_root, ignore_errors=True)
print(json.dumps(res))
