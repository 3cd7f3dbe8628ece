"""rocprofv3's kernel_stats.csv of a bench.py run mixes two kinds of launches: the frame's own kernels and the COUNTING variants of the
bench's untimed replay (`kt_tsdf23_lean_kernel<true, ...>`, `kt_raycast_kernel<true, ...>`: every wave ends with atomics on one counter
line -- 260 us "averages" that are diagnostics, never timed).  This writes the table with the counting rows moved
BELOW a separator row that says what they are, and percentages recomputed over the frame's own launches.
    python scripts/split_kernel_stats.py <in.csv> <out.csv>"""
import csv
import sys


def main(src, dst):
    rows = list(csv.reader(open(src)))
    head, body = rows[0], rows[1:]
    name = head.index("Name")
    total = head.index("TotalDurationNs")
    pct = head.index("Percentage")
    counting = [r for r in body if "<true," in r[name]]
    own = [r for r in body if "<true," not in r[name]]
    s = sum(float(r[total]) for r in own) or 1.0
    for r in own:
        r[pct] = "%.6f" % (100.0 * float(r[total]) / s)
    for r in counting:
        r[pct] = ""
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(head)
        w.writerows(own)
        note = [""] * len(head)
        note[name] = ("# BELOW: counting variants of bench.py's UNTIMED replay (U / S counters: same-address atomics at the end of every wave); "
                      "not part of any timed frame, excluded from the percentages above")
        w.writerow(note)
        w.writerows(counting)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
