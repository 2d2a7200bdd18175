#!/usr/bin/env python3
"""Golden vectors of MultiScaleBilateralAffineTransform.inverse_loss (models/modules.py:474-492), produced by IMPORTING THE REFERENCE
(build container only): the reference module's own forward() on a small image, then its inverse_loss; values and gradients w.r.t. the
per-level maps.  tests/golden/inverse_loss_*.npz

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_inverse_loss.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_neural_modules import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    M = import_reference()
    g = torch.Generator().manual_seed(11)
    for name, H, W in (("a", 12, 20), ("b", 9, 9)):
        mod = M.MultiScaleBilateralAffineTransform("Affine", n=2, grid=[[2, 2, 1], [4, 4, 2], [8, 8, 4]], device="cpu")
        with torch.no_grad():
            for p in mod.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
        rgb = torch.rand(H, W, 3, generator=g)
        infos = {"img_idx": torch.full((H, W), 1, dtype=torch.long)}
        maps = mod(rgb, infos, guidance_factor=[2, 2, 1] if H < 12 else [4, 4, 2])
        maps_leaf = [m.detach().clone().requires_grad_(True) for m in maps]
        mod.save_matrix = maps_leaf
        gt = torch.rand(H, W, 3, generator=g)
        render = torch.rand(H, W, 3, generator=g).requires_grad_(True)
        loss = mod.inverse_loss(gt, render)
        loss.backward()
        rec = {"gt": gt.numpy(), "render": render.detach().numpy(), "loss": np.array(float(loss)), "v_render": render.grad.numpy()}
        for i, m in enumerate(maps_leaf):
            rec[f"map{i}"] = m.detach().numpy()
            rec[f"v_map{i}"] = m.grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"inverse_loss_{name}.npz"), **rec)
        print(name, float(loss))


if __name__ == "__main__":
    main()
