#!/usr/bin/env python3
"""Generate golden vectors for the per-pixel image-loss terms by IMPORTING THE REFERENCE's own functions.

Runs only in the build container, where /root/reference is mounted:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_losses.py

Loads /root/reference/project/models/losses.py by file path (it imports only numpy / torch) and evaluates, exactly as
`BasicTrainer.compute_losses` does (/root/reference/project/models/trainers/base.py:518-565, loss functions built at
:230-250 from configs/omnire_ms_bilateral_extended.yaml:19-32):
    rgb_loss          = w_rgb   * |pixels*valid - rgb*valid|.mean()
    sky_loss_opacity  = w_mask  * binary_cross_entropy(opacity*valid, (1 - sky_masks)*valid, reduction="mean")
    depth_loss        = w_depth * DepthLoss(loss_type, normalize=False, use_inverse_depth=False)(depth, lidar, hit_mask)
with hit_mask = (lidar > 0) * valid, valid = 1 - egocar_masks (or ones).  Writes inputs, the three terms, their sum and
the autograd gradients w.r.t. rgb / opacity / depth to tests/golden/pixel_loss_*.npz.  Fixtures are data only.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/project/models/losses.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference_losses():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("_ref_losses", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case(ref, name, H, W, seed, depth_type, egocar, dtype):
    g = torch.Generator().manual_seed(seed)
    rgb = (torch.rand(H, W, 3, generator=g, dtype=dtype) * 1.1).requires_grad_(True)     # the transform may overshoot 1
    pixels = torch.rand(H, W, 3, generator=g, dtype=dtype)
    opacity = torch.rand(H, W, 1, generator=g, dtype=dtype)
    opacity.view(-1)[:7] = torch.tensor([0.0, 1.0, 1e-9, 1 - 1e-7, 0.5, 0.0, 1.0], dtype=dtype)   # the log clamps of BCE
    opacity.requires_grad_(True)
    sky = (torch.rand(H, W, generator=g) < 0.3).to(dtype)
    depth = (torch.rand(H, W, 1, generator=g, dtype=dtype) * 90.0)
    depth.view(-1)[7:10] = torch.tensor([0.0, 5e-5, 2e-4], dtype=dtype)                   # around the pred > 1e-4 test
    depth.requires_grad_(True)
    lidar = torch.rand(H, W, generator=g, dtype=dtype) * 100.0                             # some beyond the 80 m bound
    lidar = lidar * (torch.rand(H, W, generator=g) < 0.4).to(dtype)                        # sparse hits
    lidar.view(-1)[10:13] = torch.tensor([0.005, 0.02, 79.99], dtype=dtype)
    ego = (torch.rand(H, W, generator=g) < 0.15).to(dtype) if egocar else None
    w_rgb, w_mask, w_depth = 0.8, 0.05, 0.01
    # ---- compute_losses, line by line ----
    valid = (1.0 - ego) if ego is not None else torch.ones_like(sky)
    gt_rgb = pixels * valid[..., None]
    predicted_rgb = rgb * valid[..., None]
    gt_occupied_mask = (1.0 - sky) * valid
    pred_occupied_mask = opacity.squeeze() * valid
    Ll1 = torch.abs(gt_rgb - predicted_rgb).mean()
    rgb_loss = w_rgb * Ll1
    sky_loss = ref.binary_cross_entropy(pred_occupied_mask, gt_occupied_mask, reduction="mean") * w_mask
    lidar_hit_mask = (lidar > 0).to(dtype) * valid
    depth_fn = ref.DepthLoss(loss_type=depth_type, normalize=False, use_inverse_depth=False)
    depth_loss = depth_fn(depth, lidar, lidar_hit_mask) * w_depth
    total = rgb_loss + sky_loss + depth_loss
    total.backward()
    np.savez_compressed(
        os.path.join(OUT, f"pixel_loss_{name}.npz"), rgb=rgb.detach().numpy(), pixels=pixels.numpy(), opacity=opacity.detach().numpy(),
        sky_masks=sky.numpy(), depth=depth.detach().numpy(), lidar=lidar.numpy(), egocar=(ego.numpy() if ego is not None else np.zeros(0)),
        w=np.array([w_rgb, w_mask, w_depth]), depth_l2=np.array(1 if depth_type == "l2" else 0),
        rgb_loss=rgb_loss.detach().numpy(), sky_loss=sky_loss.detach().numpy(), depth_loss=depth_loss.detach().numpy(),
        total=total.detach().numpy(), v_rgb=rgb.grad.numpy(), v_opacity=opacity.grad.numpy(), v_depth=depth.grad.numpy())
    print(name, float(rgb_loss), float(sky_loss), float(depth_loss))


def main():
    ref = load_reference_losses()
    os.makedirs(OUT, exist_ok=True)
    case(ref, "l1_19x27_f64", 19, 27, 0, "l1", False, torch.float64)
    case(ref, "l1_19x27_f32", 19, 27, 0, "l1", False, torch.float32)
    case(ref, "l2_ego_23x31_f64", 23, 31, 1, "l2", True, torch.float64)
    case(ref, "l2_ego_23x31_f32", 23, 31, 1, "l2", True, torch.float32)
    case(ref, "l1_ego_40x64_f32", 40, 64, 2, "l1", True, torch.float32)


if __name__ == "__main__":
    main()
