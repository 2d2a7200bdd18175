#!/usr/bin/env python
"""Bilateral transform alone at the BASELINE workloads' shapes: forward / backward time (device events) with the cell-aligned
kernels (option 7 = 1) and the general kernels (0), on a noisy and on a smooth image.  usage: bilagrid_time.py [reps [shapes|all [kinds|all [cells]]]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bilateral_driving_amd import _lib as L
from bilateral_driving_amd.bilagrid import bilagrid_transform

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only_shapes = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None      # e.g. c2,c5
only_kinds = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "all" else None        # noise,smooth
only_cells = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 3]
dev = "cuda"
SHAPES = {"c2": (1080, 1920, [(16, 16, 8)], [1]),
          "headline": (1080, 1920, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
          "c3": (900, 1600, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
          "c5": (1280, 1920, [(2, 2, 1), (4, 4, 2), (8, 8, 4), (16, 16, 8)], [8, 4, 4, 2])}
g = torch.Generator().manual_seed(0)
for name, (H, W, levels, factors) in SHAPES.items():
    if only_shapes and name not in only_shapes:
        continue
    for kind in ("noise", "smooth"):
        if only_kinds and kind not in only_kinds:
            continue
        if kind == "noise":
            rgb = torch.rand(H, W, 3, generator=g)
        else:
            yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
            rgb = torch.stack([0.2 + 0.6 * xx, 0.3 + 0.4 * yy, 0.5 - 0.3 * xx * yy], dim=-1)
        rgb = rgb.to(dev).requires_grad_(True)
        alpha = torch.rand(H, W, generator=g).to(dev)
        sky = torch.rand(H, W, 3, generator=g).to(dev)
        grids = []
        for (gx, gy, gl) in levels:
            ident = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]).reshape(12, 1, 1, 1).repeat(1, gl, gy, gx)
            grids.append((ident + 0.05 * torch.randn(12, gl, gy, gx, generator=g)).to(dev).requires_grad_(True))
        wt = torch.randn(H, W, 3, generator=g).to(dev)

        def run():
            out = bilagrid_transform(rgb, grids, factors, alpha=alpha, sky=sky)
            (out * wt).sum().backward()

        for cells in only_cells:
            L.set_option(L.OPT_CELLS, cells)
            for _ in range(3):
                run()
            L.enable_timers(True)
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            t = L.timer_summary()
            L.enable_timers(False)
            print(f"{name:9s} {kind:6s} cells={cells}  " + "  ".join(f"{k} {v[1] * 1e3:7.1f} us" for k, v in sorted(t.items())), flush=True)
L.set_option(L.OPT_CELLS, 3)
