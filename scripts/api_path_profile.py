#!/usr/bin/env python3
"""The drop-in path (harness.render_view_api: the reference's call sequence over this package's operators) for a few steps at the
benchmark's size -- run under rocprofv3 --kernel-trace --stats to see what the framework's own kernels cost next to the library's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import harness as Hn
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
params = Hn.synthetic_scene(N, seed=0, device=dev)
for v in params.values(): v.requires_grad_(True)
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
sky = torch.rand(H, W, 3, generator=gen).to(dev); target = torch.rand(H, W, 3, generator=gen).to(dev)
Hn.FUSED = False
def step(i):
    v = i % len(cams)
    for p in list(params.values()) + grids: p.grad = None
    Hn.training_loss(Hn.render_view(params, cams[v], grids, v, sky), target, grids).backward()
for i in range(6): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for i in range(n): step(i)
torch.cuda.synchronize()
print(f"api path: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step = {n / (time.perf_counter() - t0):.1f} it/s")
