"""Per-frame picture of ALL streams from a rocprofv3 --kernel-trace csv of a bench.py run: for a stretch of the timed region, mean start / end of every kernel relative to
the start of the frame's first odometry launch on the main stream, per stream.  usage: stream_timeline.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
by = collections.Counter(r["Stream_Id"] for r in rows)
main = by.most_common(1)[0][0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r)) for r in rows))
# frame boundaries: starts of the odometry launch (level form) or of the first kt_icp_kernel behind a ray cast
starts = []
prev = ""
for s, e, st, n in ev:
    if st != main: continue
    if n.startswith("kt_icp_level_kernel") or (n.startswith("kt_icp_kernel") and prev.startswith("kt_raycast")): starts.append(s)
    prev = n
starts = starts[len(starts) // 5: 2 * len(starts) // 5]   # inside bench.py's timed region (the later passes are serial / counting replays)
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k in range(len(starts) - 1):
    a, b = starts[k], starts[k + 1]
    seen = collections.Counter()
    for s, e, st, n in ev:
        if s < a or s >= b: continue
        seen[(st, n)] += 1
        key = (st, n, seen[(st, n)] if not n.startswith("kt_icp_kernel") else 0)
        acc[key][0] += s - a; acc[key][1] += e - a; acc[key][2] += 1
nf = len(starts) - 1
print(f"{nf} frames, mean period {(starts[-1] - starts[0]) / nf / 1e3:.1f} us; main stream = {main}")
for (st, n, i), (s, e, c) in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
    if c < nf // 4: continue
    print(f"  stream {st:>3} {n:42s} #{i} start {s / c / 1e3:7.1f} end {e / c / 1e3:7.1f} us  ({(e - s) / c / 1e3:6.1f} us, in {c} of {nf} frames)")
