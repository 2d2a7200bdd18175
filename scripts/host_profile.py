#!/usr/bin/env python3
"""Where the HOST spends the frame loop of bench.py (cProfile over a few frames, top functions by own time): a blocking call inside
the loop -- an allocator miss, an implicit synchronisation -- stands out as one entry with ~1 ms per call.  Measurement tooling.

    python scripts/host_profile.py [--no-pipeline] [--frames 8]"""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bilateral_driving_amd import harness as Hn  # noqa: E402
from bilateral_driving_amd.dist import FlatGradients, FrameExchange  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--no-pipeline", action="store_true")
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--gaussians", type=int, default=2_000_000)
args = ap.parse_args()
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, args.gaussians
cams = Hn.ring_cameras(W, H, device=dev)
params = Hn.synthetic_scene(N, seed=0, device=dev)
for v in params.values():
    v.requires_grad_(True)
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
flat = FlatGradients(list(params.values()) + grids, sparse_rows=True)
fx = FrameExchange(flat, list(params.keys()) + [f"grid{i}" for i in range(len(grids))])
pipeline = not args.no_pipeline


def frame():
    fx.begin_frame()
    front = Hn.render_view_begin(params, cams[0]) if pipeline else None
    for v in range(len(cams)):
        skies[v].grad = None
        out = Hn.render_view(params, cams[v], grids, v, skies[v], front=front, **fx.view_kwargs(v))
        fx.begin_view(out["info"])
        loss = Hn.training_loss(out, targets[v], grids, grid_grads=fx.tail_grads())
        front = Hn.render_view_begin(params, cams[v + 1]) if pipeline and v + 1 < len(cams) else None
        loss.backward()
        fx.end_view()
    fx.end_frame()


for _ in range(3):
    frame()
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(args.frames):
    frame()
torch.cuda.synchronize()
print(f"unprofiled: {(time.perf_counter() - t0) / args.frames * 1e3:.3f} ms per frame, pipeline={pipeline}")
pr = cProfile.Profile()
pr.enable()
for _ in range(args.frames):
    frame()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
print(torch.cuda.memory_stats(dev).get("num_alloc_retries"), "alloc retries;", torch.cuda.memory_stats(dev).get("num_device_alloc"), "device allocs;",
      torch.cuda.memory_stats(dev).get("num_device_free"), "device frees; reserved GB", torch.cuda.memory_reserved(dev) / 2**30)
