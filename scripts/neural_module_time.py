#!/usr/bin/env python3
"""One 1920x1080 image through the neural bilateral variants at the shipped sizes (configs/omnire_neuralbilateral.yaml:
16x16x8 grid, 24 features; configs/omnire_ms_neuralbilateral.yaml: grids [[1,1,1],[16,16,8]], 8 features each; hidden 64):
``transform`` forward + backward (feature slice kernels + the fused MLP head), microseconds per call.  Run it under
``rocprofv3 --kernel-trace --stats`` for the per-kernel split."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bilateral_driving_amd.modules import MultiScaleNeuralBilateralAffineTransform, NeuralBilateralAffineTransform  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    H, W = 1080, 1920
    g = torch.Generator().manual_seed(0)
    # a rendered image is smooth: low-frequency content + 2 % noise (neighbouring pixels share their guidance cell, as in a real
    # frame; pure noise -- every pixel of a wave in another cell -- is the scatter's worst case, kept as the second workload)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    smooth = torch.stack([0.5 + 0.4 * torch.sin(6.0 * xx + 2.0 * yy), 0.5 + 0.4 * torch.cos(5.0 * yy - 3.0 * xx), 0.3 + 0.6 * xx * yy], -1)
    images = {"smooth": (smooth + 0.02 * torch.randn(H, W, 3, generator=g)).clamp(0, 1), "noise": torch.rand(H, W, 3, generator=g)}
    v = torch.randn(H, W, 3, generator=g).cuda()
    mods = {"single 16x16x8 f24": NeuralBilateralAffineTransform("Affine", 4, 16, 16, 8, feature_dim=24, hidden_dim=64),
            "multi [[1,1,1],[16,16,8]] f8": MultiScaleNeuralBilateralAffineTransform("Affine", 4, [[1, 1, 1], [16, 16, 8]], feature_dim=8, hidden_dim=64)}
    for name, mod in mods.items():
        with torch.no_grad():
            for p in mod.parameters():
                p.add_(torch.randn(p.shape, generator=g).cuda() * 0.1)
    for (name, mod), (iname, img) in [(m, i) for m in mods.items() for i in images.items()]:
        rgb = img.cuda().requires_grad_(True)
        infos = {"img_idx": 1}

        def fwd():
            with torch.no_grad():
                return mod.transform(rgb, infos)

        def both():
            rgb.grad = None
            for p in mod.parameters():
                p.grad = None
            (mod.transform(rgb, infos) * v).sum().backward()
        print(f"{name}, {iname} image: transform fwd {timed(fwd):8.1f} us   fwd+bwd {timed(both):8.1f} us (incl. the loss expression)")


if __name__ == "__main__":
    main()
