"""Mean per-kernel durations and inter-kernel gaps of the main stream per frame, from a rocprofv3 --kernel-trace csv.
usage: python scripts/frame_timeline.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.Counter(r['Stream_Id'] for r in rows)
main = by.most_common(1)[0][0]
rows = sorted((r for r in rows if r['Stream_Id'] == main), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][:44] for r in rows]
idx = [i for i, n in enumerate(names) if 'kt_raycast_kernel<false' in n]
acc = collections.defaultdict(float); gap_before = collections.defaultdict(float); n = 0; tot = 0
lo, hi = len(idx) // 4, 3 * len(idx) // 4
for k in range(lo, hi):
    a, b = idx[k] + 1, idx[k + 1] + 1
    pe = int(rows[a - 1]['End_Timestamp'])
    first = True
    for r, nm in zip(rows[a:b], names[a:b]):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        acc[nm] += e - s
        gap_before[("first " if first else "") + nm] += s - pe
        pe = e; first = False
    tot += int(rows[b - 1]['End_Timestamp']) - int(rows[a - 1]['End_Timestamp'])
    n += 1
print(f"frames {n}  mean frame {tot / n / 1e3:.1f} us")
for k, v in acc.items(): print(f"  {k:46s} {v / n / 1e3:8.2f} us")
print("gaps before:")
for k, v in gap_before.items():
    if v / n > 50: print(f"  {k:52s} {v / n / 1e3:8.2f} us")
