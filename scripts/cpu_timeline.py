#!/usr/bin/env python3
"""Where the HOST spends a step of the benchmark loop: time before the list-count wait, inside it, and after it.
(The wait being ~0 means the host, not the GPU, is on the critical path in front of it.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import harness as Hn, fused_view as FV

dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
params = Hn.synthetic_scene(N, seed=0, device=dev)
for v in params.values(): v.requires_grad_(True)
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
sky = torch.rand(H, W, 3, generator=gen).to(dev); target = torch.rand(H, W, 3, generator=gen).to(dev)

marks = {}
FV._host_sync_objects(dev)   # creates the pool
class EvProxy:
    def __init__(self, ev): self._ev = ev
    def __getattr__(self, k): return getattr(self._ev, k)
    def synchronize(self):
        marks["before_wait"] = time.perf_counter(); self._ev.synchronize(); marks["after_wait"] = time.perf_counter()
pool = FV._SYNC[dev]
pool["items"] = [(c, EvProxy(e)) for c, e in pool["items"]]

from bilateral_driving_amd.dist import FlatGradients, FrameExchange
flat = FlatGradients(list(params.values()) + grids, sparse_rows=True)
fx = FrameExchange(flat, list(params.keys()) + [f"grid{i}" for i in range(len(grids))])

def step(i):   # the bench's loop: one view of a frame through the arena modes
    v = i % len(cams)
    if v == 0:
        fx.begin_frame()
    marks["start"] = time.perf_counter()
    o = Hn.render_view(params, cams[v], grids, v, sky, **fx.view_kwargs(v))
    fx.begin_view(o["info"])
    marks["fwd_done"] = time.perf_counter()
    Hn.training_loss(o, target, grids, grid_grads=fx.tail_grads()).backward()
    fx.end_view()
    marks["end"] = time.perf_counter()

for i in range(6): step(i)
acc = {"pre_wait": 0, "wait": 0, "post_wait_fwd": 0, "loss_bwd": 0}
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 30
for i in range(n):
    step(i)
    acc["pre_wait"] += marks["before_wait"] - marks["start"]; acc["wait"] += marks["after_wait"] - marks["before_wait"]
    acc["post_wait_fwd"] += marks["fwd_done"] - marks["after_wait"]; acc["loss_bwd"] += marks["end"] - marks["fwd_done"]
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / n * 1e3
print(f"step {tot:.3f} ms; host: " + ", ".join(f"{k} {v / n * 1e3:.3f}" for k, v in acc.items()), f"(sum {sum(acc.values()) / n * 1e3:.3f})")
