#!/bin/bash
# HBM traffic of the GPU gzip encoder's kernels: FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, HBM: the two do not
# fit one pass; counters without any trace domain but --kernel-trace; the program right behind `--`), on tools/measure_gdeflate.py.
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 /root/repo/tools/measure_gdeflate.py --reps 3 > /tmp/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            acc[row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c + "_KiB_mean_per_launch"] = round(sum(v) / len(v), 1)
        out[k]["launches"] = len(v)
for k, v in out.items():
    if "FETCH_SIZE_KiB_mean_per_launch" in v: v["hbm_read_MB"] = round(v["FETCH_SIZE_KiB_mean_per_launch"] * 1024 * 2 / 1e6, 1)   # gfx950 tallies a 128-B request at 64 B
    if "WRITE_SIZE_KiB_mean_per_launch" in v: v["hbm_written_MB"] = round(v["WRITE_SIZE_KiB_mean_per_launch"] * 1024 / 1e6, 1)
print(json.dumps({"job": "134 MB of FASTQ text (HiFi-like qualities) in 128 members of 1 MiB", "note": "FETCH_SIZE x 1024 x 2 (reads), WRITE_SIZE x 1024 (writes); separate passes", "kernels": out}, indent=1))
PY
