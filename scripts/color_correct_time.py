#!/usr/bin/env python3
"""color_correct(rgb, pixels, 10) on one 1080p frame (models/video_utils_color_correction.py:201): the reference's formulation
(30 torch.linalg.lstsq solves of a [pixels x 10] system, bilateral/lib_bilagrid.py:94-117, written with the same torch ops on the
device) against the streaming normal-equation passes of bilateral_driving_amd.colorcorrect.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd.colorcorrect import color_correct

H, W, ITERS = 1080, 1920, 10
g = torch.Generator().manual_seed(3)
ref = (torch.rand(H, W, 3, generator=g) * 1.1 - 0.05).clamp(0, 1).cuda()
img = (ref @ (torch.eye(3) + 0.1 * torch.randn(3, 3, generator=g)).cuda().T + 0.05 * ref ** 2 - 0.02).clamp(0, 1)


def torch_formulation(img, ref, num_iters, eps=0.5 / 255):
    im, rf = img.reshape(-1, 3), ref.reshape(-1, 3)
    unc = lambda z: (z >= eps) & (z <= 1 - eps)
    mask0 = unc(im)
    for _ in range(num_iters):
        a = torch.cat([im[:, c:c + 1] * im[:, c:] for c in range(3)] + [im, torch.ones_like(im[:, :1])], dim=-1)
        warp = []
        for c in range(3):
            b = rf[:, c]
            m = mask0[:, c] & unc(im[:, c]) & unc(b)
            warp.append(torch.linalg.lstsq(torch.where(m[:, None], a, torch.zeros_like(a)), torch.where(m, b, torch.zeros_like(b))[:, None]).solution[:, 0])
        im = torch.clip(a @ torch.stack(warp, dim=-1), 0, 1)
    return im.reshape(img.shape)


for name, fn in (("torch lstsq formulation", torch_formulation), ("streaming normal equations", color_correct)) * 2:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn(img, ref, ITERS)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"{name:28s} {dt:9.2f} ms per 1080p frame   mean |out - ref| = {float((out - ref).abs().mean()):.5f}")
