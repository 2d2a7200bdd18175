#!/usr/bin/env python3
"""Golden vectors for the FEATURE-grid slice (first slice) by IMPORTING THE REFERENCE:
bilateral/lib_bilagrid.py NeuralBilateralGrid (:370-461), slice_feature (:232-253), total_variation_loss (:152-168).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_neural.py        (build container only; tensorly stubbed)

Each file holds inputs, the sliced features, the TV value and the autograd gradients w.r.t. the grids and the guiding rgb
for loss = sum(features * w) + 0.3 * tv."""
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    m = types.ModuleType("tensorly"); m.set_backend = lambda *_: None
    sys.modules["tensorly"] = m
    sys.path.insert(0, "/root/reference/project")
    import bilateral.lib_bilagrid as LB
    return LB


def case(LB, name, num, gx, gy, gl, f, H, W, dtype, multi):
    g = torch.Generator().manual_seed(len(name) + f)
    net = LB.NeuralBilateralGrid(num, gx, gy, gl, feature_dim=f).to(dtype)
    with torch.no_grad():
        net.grids.copy_(torch.randn(net.grids.shape, generator=g, dtype=dtype) * 0.3)
    net.rgb2gray_weight = net.rgb2gray_weight.to(dtype)
    net.rgb2gray = lambda rgb: (rgb @ net.rgb2gray_weight.T) * 2.0 - 1.0
    if multi:   # pixels from several views: leading dim = batch, one grid index per entry
        B = 4
        xy = torch.rand(B, H, W, 2, generator=g, dtype=dtype)
        rgb = (torch.rand(B, H, W, 3, generator=g, dtype=dtype) * 1.3 - 0.15).requires_grad_(True)
        idx = torch.tensor([2, 0, 2, 1])[:, None, None, None].expand(B, H, W, 1)
    else:
        ys, xs = torch.meshgrid(torch.linspace(0, 1, H, dtype=dtype), torch.linspace(0, 1, W, dtype=dtype), indexing="ij")
        xy = torch.stack([xs, ys], -1)
        rgb = (torch.rand(H, W, 3, generator=g, dtype=dtype) * 1.3 - 0.15).requires_grad_(True)
        idx = torch.full((H, W, 1), 1, dtype=torch.long)
    feats = LB.slice_feature(net, xy, rgb, idx)["affine_features"]
    tv = net.tv_loss()
    w = torch.randn(feats.shape, generator=g, dtype=dtype)
    ((feats * w).sum() + 0.3 * tv).backward()
    np.savez_compressed(os.path.join(OUT, f"neural_slice_{name}.npz"), grids=net.grids.detach().numpy(), xy=xy.numpy(),
                        rgb=rgb.detach().numpy(), idx=idx.numpy(), w=w.numpy(), feats=feats.detach().numpy(), tv=tv.detach().numpy(),
                        v_grids=net.grids.grad.numpy(), v_rgb=rgb.grad.numpy())
    print(name, tuple(feats.shape), float(tv))


def main():
    LB = import_reference()
    case(LB, "f8_single_f64", 3, 5, 4, 3, 8, 9, 13, torch.float64, False)
    case(LB, "f8_single_f32", 3, 5, 4, 3, 8, 9, 13, torch.float32, False)
    case(LB, "f24_single_f32", 2, 8, 8, 4, 24, 12, 10, torch.float32, False)
    case(LB, "f8_multi_f32", 3, 4, 4, 2, 8, 5, 7, torch.float32, True)
    case(LB, "f13_L1_f32", 2, 3, 3, 1, 13, 6, 6, torch.float32, False)


if __name__ == "__main__":
    main()
