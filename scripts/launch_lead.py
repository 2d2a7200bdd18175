#!/usr/bin/env python3
"""How far the HOST is ahead of the GPU at every kernel launch of one view of the bench: joins rocprofv3's HIP API trace with its
kernel trace on the correlation id and prints, per kernel, when the host issued the launch and when the GPU started it (relative to
the view's first kernel).  lead = GPU start - host launch: a few microseconds means the GPU was waiting for the host there.
Measurement tooling, run on the GPU box (the raw traces are too large to travel back).

    rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 2 --warmup 2 ...
    python scripts/launch_lead.py out/<...>/t_hip_api_trace.csv out/<...>/t_kernel_trace.csv > launch_lead.txt"""
import csv
import sys


def main():
    api_path, ker_path = sys.argv[1], sys.argv[2]
    launch = {}
    with open(api_path) as f:
        for r in csv.DictReader(f):
            fn = r.get("Function", "")
            if "Launch" in fn or "launch" in fn:
                launch[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fn)
    rows = []
    with open(ker_path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Correlation_Id"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "project_view_fwd_kernel" in r[2]]
    if len(marks) < 4:
        print("not enough views", len(marks))
        return
    a, b = marks[-3], marks[-2]
    t0 = rows[a][0]
    prev_end = t0
    print(f"{'gpu_start':>9} {'dur':>7} {'gap':>6} {'host_launch':>11} {'lead':>8}  kernel   (us; lead = gpu_start - host_launch)")
    starving = 0.0
    for s, e, name, cid in rows[a:b]:
        gap = max(0, s - prev_end) / 1e3
        la = launch.get(cid)
        host = (la[1] - t0) / 1e3 if la else float("nan")
        lead = (s - la[1]) / 1e3 if la else float("nan")
        if la and gap > 2.0 and lead < 15.0:
            starving += gap
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:6.1f} {host:11.1f} {lead:8.1f}  {name.replace('void ', '')[:70]}")
        prev_end = max(prev_end, e)
    print(f"# view window {(rows[b][0] - t0) / 1e3:.1f} us; idle gaps in front of kernels launched < 15 us before they started (host-bound): {starving:.1f} us")


if __name__ == "__main__":
    main()
