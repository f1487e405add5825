import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
print(files)
rows = list(csv.DictReader(open(files[0])))
print(rows[0].keys())
K = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:40]
    K[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r.get("Stream_Id")))
t0 = min(v[0][0] for v in K.values())
for name, v in sorted(K.items(), key=lambda kv: -sum(e - s for s, e, *_ in kv[1])):
    d = sorted(e - s for s, e, *_ in v)
    print(f"{name:40s} n={len(v):6d} total={sum(d)/1e6:9.1f} ms  median={d[len(d)//2]/1e3:9.1f} us  p90={d[int(len(d)*.9)]/1e3:9.1f} us max={d[-1]/1e3:9.1f} us queues={sorted(set(q for *_, q, st in v))}")
# overlap of probe kernels with inflate kernels
infl = sorted((s, e) for s, e, *_ in K.get("gi_inflate_kernel", []))
def overlap(s, e):
    tot = 0
    for a, b in infl:
        if b <= s: continue
        if a >= e: break
        tot += min(e, b) - max(s, a)
    return tot
for name in K:
    if "probe" in name or "count" in name or "unpack" in name:
        v = K[name]
        inside = [(e - s, overlap(s, e)) for s, e, *_ in v]
        beside = [d for d, o in inside if o > 0.5 * d]
        alone = [d for d, o in inside if o <= 0.5 * d]
        med = lambda x: sorted(x)[len(x)//2] / 1e3 if x else 0
        print(f"{name}: {len(beside)} beside an inflate kernel (median {med(beside):.0f} us), {len(alone)} alone (median {med(alone):.0f} us)")
span = max(e for v in K.values() for s, e, *_ in v) - t0
print("span of the trace %.2f s; inflate kernels busy %.2f s" % (span / 1e9, sum(e - s for s, e in infl) / 1e9))
# resources per kernel, and where the probe kernels start relative to the inflate kernels
seen = set()
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:40]
    if name in seen: continue
    seen.add(name)
    print(f"{name:40s} wg={r['Workgroup_Size_X']:>5s} grid={r['Grid_Size_X']:>9s} lds={r['LDS_Block_Size']:>6s} scratch={r['Scratch_Size']:>5s} vgpr={r['VGPR_Count']:>4s} agpr={r['Accum_VGPR_Count']:>4s} sgpr={r['SGPR_Count']:>4s}")
probes = sorted((s, e) for n, v in K.items() if "probe" in n and "false, fa" in n for s, e, *_ in v)
import bisect
ends = [e for s, e in infl]
starts = [s for s, e in infl]
after = []
for s, e in probes:
    i = bisect.bisect_right(ends, s) - 1
    j = bisect.bisect_right(starts, s) - 1
    running = j >= 0 and infl[j][1] > s
    after.append(((s - ends[i]) / 1e6 if i >= 0 else -1, running))
print("probe kernels (the main pass): ms after the last inflate kernel's end -> [running beside]:", " ".join(f"{a:.1f}{'*' if r else ''}" for a, r in after[:80]))
gaps = [(infl[i + 1][0] - infl[i][1]) / 1e6 for i in range(len(infl) - 1)]
print("gaps between inflate kernels (ms):", " ".join(f"{g:.1f}" for g in gaps))
