#!/usr/bin/env python3
"""Time of the fused neural head (csrc/mlp_head.hip) against the framework's Linear / Tanh / matmul modules on one 1920x1080 image
(the reference's form: models/modules.py:621-627 + trainers/scene_graph.py:99-106).  Prints microseconds per call."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bilateral_driving_amd import mlp_head  # noqa: E402


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
    P = 1920 * 1080
    for F in (24, 16):
        g = torch.Generator().manual_seed(0)
        mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).cuda().requires_grad_(True)
        feats, w1, w2, w3 = mk(P, F), mk(64, F), mk(64, 64), mk(12, 64)
        rgb = torch.rand(P, 3, generator=g).cuda().requires_grad_(True)
        v = torch.randn(P, 3, generator=g).cuda()

        def fused_fwd():
            with torch.no_grad():
                return mlp_head.transform(feats, rgb, w1, w2, w3)

        def fused_both():
            for t in (feats, rgb, w1, w2, w3):
                t.grad = None
            (mlp_head.transform(feats, rgb, w1, w2, w3) * v).sum().backward()

        def torch_head():
            A = (torch.tanh(torch.tanh(feats @ w1.T) @ w2.T) @ w3.T).reshape(-1, 3, 4)
            return (A[..., :3] @ rgb[..., None])[..., 0] + A[..., 3] + rgb

        def torch_fwd():
            with torch.no_grad():
                return torch_head()

        def torch_both():
            for t in (feats, rgb, w1, w2, w3):
                t.grad = None
            (torch_head() * v).sum().backward()

        flop_f = 2 * (64 * F + 4096 + 768) * P
        tf, tb = timed(fused_fwd), timed(fused_both)
        rf, rb = timed(torch_fwd), timed(torch_both, n=5)
        print(f"F={F}: fused fwd {tf:8.1f} us ({flop_f / tf / 1e6:6.1f} TFLOP/s useful)  fwd+bwd {tb:8.1f} us ({4 * flop_f / tb / 1e6:6.1f} TFLOP/s useful; "
              f"includes the loss expression) | framework fwd {rf:8.1f} us  fwd+bwd {rb:8.1f} us | speed-up fwd {rf / tf:4.1f}x  fwd+bwd {rb / tb:4.1f}x")


if __name__ == "__main__":
    main()
