#!/usr/bin/env python3
"""Golden vectors from the few geometry / colour helpers the REFERENCE TREE itself holds for the rasterizer half (whose arithmetic
otherwise lives in gsplat, absent): /root/reference/project/utils/geometry.py::transform_points (:7-21) and
project_camera_points_to_image (:39-57) -- the pinhole convention the dataset side projects lidar points with -- and
models/gaussians/basics.py::SH2RGB / RGB2SH (:76-89), the degree-0 SH constant.  They anchor the oracle's world -> camera transform,
its means2d / depths and its degree-0 colour; everything else of oracle/gs_oracle.py stays parity-unpinned.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_geometry.py        (build container only)"""
import importlib.util
import os

import numpy as np
import torch

import gen_golden_refine as G


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    G.import_reference()
    geo = load("ref_geometry", "/root/reference/project/utils/geometry.py")
    from models.gaussians.basics import RGB2SH, SH2RGB
    g = torch.Generator().manual_seed(11)
    N = 200
    pts = torch.randn(N, 3, generator=g, dtype=torch.float64) * 3 + torch.tensor([0.0, 0.0, 9.0], dtype=torch.float64)
    ang = 0.3
    R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float64)
    c2w = torch.eye(4, dtype=torch.float64); c2w[:3, :3] = R; c2w[:3, 3] = torch.tensor([0.5, -0.2, 1.0], dtype=torch.float64)
    w2c = torch.linalg.inv(c2w)
    K = torch.tensor([[700.0, 0, 320.0], [0, 710.0, 180.0], [0, 0, 1]], dtype=torch.float64)
    cam = geo.transform_points(pts, w2c)
    uv, depth = geo.project_camera_points_to_image(cam.numpy(), K.numpy())
    sh = torch.randn(N, 3, generator=g, dtype=torch.float64)
    rec = {"points": pts.numpy(), "c2w": c2w.numpy(), "K": K.numpy(), "cam_points": cam.numpy(), "uv": uv, "depth": depth,
           "sh_dc": sh.numpy(), "rgb_from_sh": SH2RGB(sh).numpy(), "sh_from_rgb": RGB2SH(SH2RGB(sh)).numpy()}
    np.savez_compressed(os.path.join(G.OUT, "ref_geometry.npz"), **rec)
    print("ok", uv.shape, float(depth.min()), float(depth.max()))


if __name__ == "__main__":
    main()
