"""What ran BESIDE a kernel: from a rocprofv3 --kernel-trace csv, for every launch of the target kernel (default: the voxel kernel's timed
variant) the kernels of OTHER streams whose [start, end) intersects it, with the overlap in us.  usage: overlap_report.py <kernel_trace.csv> [substring]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "kt_tsdf23_lean_kernel<false"
short = lambda r: r["Kernel_Name"].split("(")[0][:48]
iv = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")), short(r)) for r in rows))
tg = [x for x in iv if key in x[3]]
tg = tg[len(tg) // 4:]   # the steady part of the run
beside = collections.defaultdict(float)
dur = 0.0
clean = []
for s, e, st, n in tg:
    dur += e - s
    ov = 0.0
    for s2, e2, st2, n2 in iv:
        if e2 <= s or s2 >= e or (s2 == s and e2 == e and n2 == n):
            continue
        o = min(e, e2) - max(s, s2)
        beside[n2 + " [stream %s]" % st2] += o
        ov += o
    clean.append((e - s, ov))
print(f"{len(tg)} launches of '{key}', mean {dur / max(1, len(tg)) / 1e3:.1f} us")
alone = [d for d, o in clean if o < 0.02 * d]
shared = [d for d, o in clean if o >= 0.02 * d]
if alone: print(f"  {len(alone)} launches with < 2 % overlap: mean {sum(alone) / len(alone) / 1e3:.1f} us")
if shared: print(f"  {len(shared)} launches with overlap: mean {sum(shared) / len(shared) / 1e3:.1f} us")
for k, v in sorted(beside.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  beside: {k:70s} {v / max(1, len(tg)) / 1e3:8.2f} us per launch")
