#!/usr/bin/env python3
"""Optimiser step on the benchmark's parameter set (2 M Gaussians x 59 floats): torch.optim.Adam vs FusedAdam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import harness as Hn
from bilateral_driving_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
for cls in (torch.optim.Adam, FusedAdam, torch.optim.Adam, FusedAdam):
    params = Hn.synthetic_scene(2_000_000, seed=0, device=dev)
    ps = [v.requires_grad_(True) for v in params.values()]
    opt = cls([{"params": [p], "lr": 1e-3, "eps": 1e-15, "weight_decay": 0} for p in ps], lr=0.0, eps=1e-15)
    for p in ps: p.grad = torch.randn_like(p)
    for _ in range(3): opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): opt.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    n = sum(p.numel() for p in ps)
    print(f"{cls.__name__:10s} {dt:.3f} ms/step  ({28 * n / dt / 1e6:.0f} GB/s of the 28 B/element minimum)")
