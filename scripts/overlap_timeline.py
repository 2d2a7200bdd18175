#!/usr/bin/env python3
"""The REAL timeline of a graph-replayed, two-stream frame (rocprofv3 serialises the streams, so its trace cannot show it): every
operator of every view is bracketed by timing marks that are event-record nodes of the captured graphs (bds_timer_mark); after a
replay the marks' timestamps, relative to the frame's first mark, give who ran next to whom.

    python scripts/overlap_timeline.py [--no-overlap]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
from bilateral_driving_amd.graph_view import FrameGraph

dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
for c in cams:
    c.viewmat.requires_grad_(True)
params = {k: v.requires_grad_(True) for k, v in Hn.synthetic_scene(N, seed=0, device=dev).items()}
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
L.enable_timers(True)
frame = FrameGraph(params, cams, grids, skies, targets, overlap="--no-overlap" not in sys.argv)
L.enable_timers(False)
for _ in range(4):
    frame.step(wait=False)
torch.cuda.synchronize()
lib = L.lib()
ref = frame.marks["project_fwd"][0][0]
rows = []
for name, pairs in frame.marks.items():
    for v, (s, e) in enumerate(pairs):
        a, b = lib.bds_timer_elapsed_ms(ref.h, s.h), lib.bds_timer_elapsed_ms(ref.h, e.h)
        rows.append((a * 1e3, b * 1e3, name, v))
rows.sort()
fwd = {"project_fwd", "isect_prepare", "sh_fwd", "isect_build", "rasterize_fwd", "bilagrid_fwd"}
print(f"# frame: {max(r[1] for r in rows):.0f} us from the first mark to the last; columns: start_us end_us dur_us stream op view")
for a, b, name, v in rows:
    print(f"{a:8.0f} {b:8.0f} {b - a:7.0f}  {'S2 fwd' if name in fwd else 'S1 bwd'}  {name:14s} v{v}")
