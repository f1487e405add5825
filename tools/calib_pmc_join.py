"""Joins tools/gpu_r5_c.sh's PMC passes: per kernel, the mean counter values per launch (full-size launches only)."""
import collections, csv, glob, json, os, sys
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
res = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "cal_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[4:]
    variant = next((v for v in ("filter_lut", "noprompt", "prompt", "filter", "starts", "base") if tag.startswith(v + "_")), None)
    group = "bench_" + variant if variant else "calib"
    rows = []
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        rows += list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for (kern, ctr), vals in by.items():
        if "calib" not in kern and "probe_entry_kernel" not in kern:
            continue
        full = max(g for g, _ in vals)
        v = [x for g, x in vals if g == full]
        res[group + ":" + kern][ctr] = round(sum(v) / len(v))
        res[group + ":" + kern]["launches_" + ctr] = len(v)
for logf in glob.glob(os.path.join(out, "cal_TCC_EA0_RDREQ_sum*.log")):
    for line in open(logf):
        if line.startswith('{"footprint'):
            res["calib_lines"] = json.loads(line)
print(json.dumps(res, indent=1))
