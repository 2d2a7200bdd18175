#!/usr/bin/env python3
"""Interleaved A/B of one bds_set_option switch on the benchmark step:  ab_option.py <which> <value_a> <value_b> [timer]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
which, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tname = sys.argv[4] if len(sys.argv) > 4 else "isect_build"
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
    o = Hn.render_view(params, cams[v], grids, v, sky)
    Hn.training_loss(o, target, grids).backward()
for i in range(6): step(i)
res = {va: [], vb: []}
for rnd in range(4):
    for val in (va, vb):
        L.set_option(which, val)
        L.enable_timers(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(12): step(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12 * 1e3
        ts = L.timer_summary(); L.enable_timers(False)
        res[val].append((dt, ts[tname][1]))
for val in (va, vb):
    print(f"option {which} = {val}  step ms:", " ".join(f"{a:.3f}" for a, _ in res[val]), f"  {tname} ms:", " ".join(f"{b:.4f}" for _, b in res[val]))
