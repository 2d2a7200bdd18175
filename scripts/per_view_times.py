#!/usr/bin/env python3
"""Per-VIEW operator times of a replayed frame (in-graph marks, one stream): which camera of the rig costs what.

    python scripts/per_view_times.py [--scene ring|lidar] [--gaussians N]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
from bilateral_driving_amd.graph_view import FrameGraph

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="lidar")
ap.add_argument("--gaussians", type=int, default=None)
a = ap.parse_args()
dev = torch.device("cuda", 0)
W, H = 1920, 1080
N = a.gaussians or (1_000_000 if a.scene == "lidar" else 2_000_000)
cams = Hn.ring_cameras(W, H, device=dev)
for c in cams:
    c.viewmat.requires_grad_(True)
scene = Hn.lidar_scene(N, device=dev) if a.scene == "lidar" else Hn.synthetic_scene(N, device=dev)
params = {k: v.requires_grad_(True) for k, v in scene.items()}
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
L.enable_timers(True)
frame = FrameGraph(params, cams, grids, skies, targets)
L.enable_timers(False)
for _ in range(3):
    frame.step(serial=True)
torch.cuda.synchronize()
print("counts (M, visible) per view:", frame.counts(), "split_len per view:", frame.split_len)
with torch.no_grad():
    for v, cam in enumerate(cams):
        info = Hn.render_view(params, cam, grids, v, skies[v])["info"]
        offs = info["isect_offsets"].reshape(-1).long()
        lens = torch.diff(offs, append=offs.new_tensor([int(info["n_isects"])])).float()
        print(f"view {v}: list-tile lists: mean {float(lens.mean()):.0f} max {int(lens.max())} q99 {float(lens.quantile(0.99)):.0f} lists >= 6144: {int((lens >= 6144).sum())}")
for name in sorted(frame.marks):
    print(f"{name:16s}", " ".join(f"{ms * 1e3:7.0f}" for ms in frame.mark_samples(name)), "us per view")
