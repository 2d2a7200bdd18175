#!/usr/bin/env python3
"""Two-stream frame from a rocprofv3 --kernel-trace CSV: for the last complete graph-replayed frames, the wall time of a frame,
the time at least one kernel runs, the time two or more run, and every kernel's mean in-situ duration next to its share.

    python scripts/frame_overlap_report.py <bench_kernel_trace.csv> [frames_from_end_to_skip=2] [frames=3]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a frame starts with the row-wise gradient clears of the begin stage (6 launches within the first ~200 us of a frame, since the
    # begin stage was split in two graphs interleaved with the first forward's kernels): a clear more than 1 ms after the previous one
    marks, last = [], None
    for i, r in enumerate(rows):
        if "view_grads_clear_list_kernel" in r[2]:
            if last is None or r[0] - last > 1_000_000:
                marks.append(i)
            last = r[0]
    if len(marks) < 3:
        print("not enough frames in the trace", len(marks)); return
    # frames = mark-to-mark windows; the graph-replayed ones are the shortest and the most numerous: keep those within 10 % of the minimum
    wins = [(rows[marks[i + 1]][0] - rows[marks[i]][0], marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
    wmin = min(w for w, _, _ in wins)
    sel = [(w, a, b) for w, a, b in wins if w <= 1.1 * wmin]
    nfr = len(sel)
    busy1 = busy2 = tot = 0
    per = defaultdict(lambda: [0, 0])
    for w, a, b in sel:
        t0, t1 = rows[a][0], rows[b][0]
        ev = []
        for s, e, n in rows[a:b]:
            ev.append((s, 1)); ev.append((min(e, t1), -1))
            k = n.replace("void ", "").split("(")[0][:70]
            per[k][0] += 1; per[k][1] += e - s
        ev.sort()
        depth, last = 0, t0
        for t, d in ev:
            if depth >= 1: busy1 += t - last
            if depth >= 2: busy2 += t - last
            depth += d; last = t
        tot += t1 - t0
    wall = tot / nfr / 1e3
    print(f"# {nfr} graph-replayed frames of {len(wins)} in the trace: {wall:.1f} us per frame; some kernel running {busy1 / tot:.3f} of the time, "
          f"two or more {busy2 / tot:.3f}")
    print(f"# sum of kernel durations per frame {sum(v[1] for v in per.values()) / nfr / 1e3:.1f} us")
    print(f"{'calls/frame':>11} {'mean_us':>8} {'us/frame':>9}  kernel")
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{c / nfr:11.1f} {t / c / 1e3:8.1f} {t / nfr / 1e3:9.1f}  {k}")


if __name__ == "__main__":
    main()
