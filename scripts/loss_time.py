#!/usr/bin/env python3
"""Image loss of one 1080p training step (rgb L1 + SSIM + sky-mask BCE + lidar-depth L1, forward + backward):
framework-op formulation (what the reference trainer runs; SSIM through F.conv2d as pytorch_msssim does) vs the HIP ops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bilateral_driving_amd.losses import pixel_loss, ssim_loss
from oracle import loss_oracle as LO   # measurement script only: the framework formulation is the oracle's restatement

dev = torch.device("cuda", 0)
H, W = 1080, 1920
g = torch.Generator().manual_seed(0)
rgb = torch.rand(H, W, 3, generator=g).to(dev).requires_grad_(True)
opacity = torch.rand(H, W, 1, generator=g).to(dev).requires_grad_(True)
depth = (torch.rand(H, W, 1, generator=g) * 90).to(dev).requires_grad_(True)
pixels = torch.rand(H, W, 3, generator=g).to(dev)
sky = (torch.rand(H, W, generator=g) < 0.3).float().to(dev)
lidar = (torch.rand(H, W, generator=g) * 100 * (torch.rand(H, W, generator=g) < 0.4)).to(dev)


def framework():
    valid = torch.ones_like(sky)
    l1 = torch.abs(pixels * valid[..., None] - rgb * valid[..., None]).mean()
    sim = 1 - LO.ssim(pixels, rgb)
    bce = F.binary_cross_entropy(opacity.squeeze() * valid, (1.0 - sky) * valid, reduction="none").mean()
    hit = (lidar > 0).float() * valid
    pd, gt = depth.squeeze() * hit, lidar * hit
    m = (gt > 0.01) & (gt < 80) & (pd > 0.0001)
    dl = F.l1_loss(pd[m], gt[m], reduction="none").mean()
    return 0.8 * l1 + 0.2 * sim + 0.05 * bce + 0.01 * dl


def fused():
    return pixel_loss(rgb, opacity, depth, pixels, sky, lidar).sum() + 0.2 * ssim_loss(rgb, pixels)


for name, fn in (("framework", framework), ("fused", fused), ("framework", framework), ("fused", fused)):
    for _ in range(3):
        for t in (rgb, opacity, depth): t.grad = None
        fn().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        for t in (rgb, opacity, depth): t.grad = None
        loss = fn(); loss.backward()
    torch.cuda.synchronize()
    print(f"{name:10s} {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms  loss {float(loss):.6f}")
