#!/usr/bin/env python3
"""Cube-map sky (EnvLight, resolution 1024 as in the shipped configs) at 1920x1080: forward and texture-gradient kernels.
Run on the GPU box:  python scripts/envlight_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import harness as Hn, _lib as L
from bilateral_driving_amd.envlight import EnvLight

W, H = 1920, 1080
cam = Hn.ring_cameras(W, H, device="cuda")[0]
jj, ii = torch.meshgrid(torch.arange(W, device="cuda") + 0.5, torch.arange(H, device="cuda") + 0.5, indexing="xy")
dirs = torch.nn.functional.normalize(torch.stack([jj, ii, torch.ones_like(jj)], -1) @ torch.linalg.inv(cam.K).T
                                     @ torch.linalg.inv(cam.viewmat)[:3, :3].T, dim=-1).contiguous()
sky = EnvLight("Sky", resolution=1024)
with torch.no_grad():
    sky.base.copy_(torch.rand_like(sky.base))
px = H * W
rows = torch.arange(H, device="cuda")[:, None, None]
masks = {"sky = upper 30 % of the image (a street scene)": (rows < 0.3 * H).float().expand(H, W, 1),
         "sky = random 30 % of the pixels (worst case)": (torch.rand(H, W, 1, device="cuda") < 0.3).float()}
for name, m in masks.items():
    v = (torch.rand(H, W, 3, device="cuda") * m).contiguous()
    for flat in (False, True):
        d = dirs.reshape(-1, 3) if flat else dirs
        vv = v.reshape(-1, 3) if flat else v
        for _ in range(3):
            sky.base.grad = None
            sky({"viewdirs": d}).backward(vv)
        L.enable_timers(True)
        for _ in range(20):
            sky.base.grad = None
            sky({"viewdirs": d}).backward(vv)
        torch.cuda.synchronize()
        t = L.timer_summary()
        L.enable_timers(False)
        print(f"{name}, {'one atomic per tap' if flat else '16x16 tiles pre-summed in LDS'}: fwd {t['cubemap_fwd'][1] * 1e3:.1f} us  "
              f"bwd {t['cubemap_bwd'][1] * 1e3:.1f} us per 1080p view  (fwd {px * 24 / t['cubemap_fwd'][1] / 1e6:.0f} GB/s of directions + colours)")
