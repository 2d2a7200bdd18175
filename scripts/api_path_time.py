#!/usr/bin/env python3
"""Step time of the benchmark scene through the reference-signature operators (rasterization(), spherical_harmonics(),
MultiScaleBilateralAffineTransform-style transform) versus the fused single-node view."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
params = Hn.synthetic_scene(N, seed=0, device=dev)
for v in params.values(): v.requires_grad_(True)
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
sky = torch.rand(H, W, 3, generator=gen).to(dev); target = torch.rand(H, W, 3, generator=gen).to(dev)
def step(i):
    v = i % len(cams)
    for p in list(params.values()) + grids: p.grad = None
    o = Hn.render_view(params, cams[v], grids, v, sky)
    Hn.training_loss(o, target, grids).backward()
for fused in (True, False, True, False):
    Hn.FUSED = fused
    for i in range(6): step(i)
    L.enable_timers(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(12): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12 * 1e3
    ts = L.timer_summary(); L.enable_timers(False)
    print("fused" if fused else "api  ", f"{dt:.3f} ms/step", {k: round(v[1], 3) for k, v in sorted(ts.items())})
Hn.FUSED = True
