#!/usr/bin/env python3
"""One view of the bench, kernel by kernel, from a rocprofv3 --kernel-trace CSV: which kernels run between two consecutive
projection launches, how long each takes and how long the GPU idles in front of it.  Written on the GPU box (the full trace is too
large to travel back); the small text table goes under profiles/.

    python scripts/step_timeline.py <bench_kernel_trace.csv> [view_index_from_end=2] > timeline.txt"""
import csv
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "project_view_fwd_kernel" in r[2]]
    if len(marks) < back + 1:
        print("not enough views in the trace", len(marks))
        return
    a, b = marks[-back - 1], marks[-back]
    t0 = rows[a][0]
    prev_end = t0
    busy = {"bds": 0, "torch": 0, "copy": 0}
    idle = 0
    print(f"# view window: {(rows[b][0] - t0) / 1e3:.1f} us, {b - a} kernels")
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel")
    for s, e, name in rows[a:b]:
        gap = max(0, s - prev_end)
        idle += gap
        kind = "bds" if "bds::" in name else ("copy" if "rocclr" in name else "torch")
        busy[kind] += e - s
        short = name.replace("void ", "")
        short = short[:100]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:7.1f}  {short}")
        prev_end = max(prev_end, e)
    tail = max(0, rows[b][0] - prev_end)
    idle += tail
    print(f"# busy: bds {busy['bds'] / 1e3:.1f} us, torch {busy['torch'] / 1e3:.1f} us, copies/fills {busy['copy'] / 1e3:.1f} us; idle {idle / 1e3:.1f} us "
          f"(incl. {tail / 1e3:.1f} us before the next view)")


if __name__ == "__main__":
    main()
