"""Shared helpers for the tests (synthetic scenes, error metrics, host-math shim loader)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_scene(N, W, H, seed=0, dtype=torch.float32, device="cpu", fov_deg=70.0, spread=1.0):
    """Small synthetic scene in front of one camera (OpenCV convention, z forward)."""
    g = torch.Generator().manual_seed(seed)
    fx = 0.5 * W / np.tan(np.deg2rad(fov_deg) / 2)
    z = torch.rand(N, generator=g, dtype=torch.float64) * 8 + 2.0
    x = (torch.rand(N, generator=g, dtype=torch.float64) - 0.5) * 1.4 * z * (W / fx) * spread
    y = (torch.rand(N, generator=g, dtype=torch.float64) - 0.5) * 1.4 * z * (H / fx) * spread
    means = torch.stack([x, y, z], -1)
    quats = torch.randn(N, 4, generator=g, dtype=torch.float64)
    # per-axis pixel-space sigma log-uniform in [0.7, W/8] px -> world scale at depth z
    spx = torch.exp(torch.rand(N, 3, generator=g, dtype=torch.float64) * np.log((W / 8) / 0.7) + np.log(0.7))
    scales = spx * z[:, None] / fx
    opac = torch.sigmoid(torch.randn(N, generator=g, dtype=torch.float64) * 1.5)
    colors = torch.rand(N, 3, generator=g, dtype=torch.float64)
    # camera: small rotation + translation so that viewmat is not the identity
    ang = 0.1
    Rz = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=torch.float64)
    Ry = torch.tensor([[np.cos(0.05), 0, np.sin(0.05)], [0, 1, 0], [-np.sin(0.05), 0, np.cos(0.05)]], dtype=torch.float64)
    vm = torch.eye(4, dtype=torch.float64)
    vm[:3, :3] = Rz @ Ry
    vm[:3, 3] = torch.tensor([0.1, -0.05, 0.2], dtype=torch.float64)
    # move the Gaussians into world space so that they land in front of the camera
    means = (means - vm[:3, 3]) @ vm[:3, :3]
    K = torch.tensor([[fx, 0, W / 2], [0, fx, H / 2], [0, 0, 1]], dtype=torch.float64)
    out = dict(means=means, quats=quats, scales=scales, opacities=opac, colors=colors, viewmats=vm[None], Ks=K[None])
    return {k: v.to(dtype).to(device) for k, v in out.items()}


def grad_errors(got, ref, floor=1e-3):
    """(norm-relative error, largest and 99th-percentile ELEMENT-wise relative error over the elements whose reference magnitude is
    above `floor` x the largest one -- below that a gradient entry is a sum of cancelling terms and its relative error says nothing)."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    nrm = float((got - ref).norm() / ref.norm().clamp(min=1e-300))
    big = ref.abs() > floor * ref.abs().max()
    if not bool(big.any()):
        return nrm, 0.0, 0.0
    e = ((got - ref).abs() / ref.abs().clamp(min=1e-300))[big]
    return nrm, float(e.max()), float(torch.quantile(e, 0.99)) if e.numel() < 16_000_000 else float(e.kthvalue(int(0.99 * e.numel())).values)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp(min=1e-12))


_HM = None


def hostmath():
    """Build (hipcc, host side only) and load the test-only shim around the kernels' HD math."""
    global _HM
    if _HM is not None:
        return _HM
    src = os.path.join(ROOT, "tests", "hostmath_shim.hip")
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libhostmath.so")
    deps = [src, os.path.join(ROOT, "bilateral_driving_amd", "csrc", "gs_math.h"),
            os.path.join(ROOT, "bilateral_driving_amd", "csrc", "bilagrid_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        hipcc = "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so])
    _HM = ctypes.CDLL(so)
    return _HM


def fptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)
