#!/usr/bin/env python3
"""Generate golden vectors for the bilateral-grid half by IMPORTING THE REFERENCE.

Runs only in the build container, where /root/reference is mounted:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_bilateral.py

Writes small ``.npz`` fixtures (inputs + the reference's outputs and autograd gradients)
into ``tests/golden/``.  The fixtures are data; no reference source is copied.  Three
imports the bilateral code never touches are stubbed (tensorly, pytorch3d.ops,
nvdiffrast.torch) so that ``models.modules`` can be imported on a CPU-only box
(SURVEY.md appendix D).

Reference entry points exercised (under /root/reference/project):
  models/modules.py:422-593   MultiScaleBilateralAffineTransform (train + test branch)
  models/modules.py:275-351   BilateralAffineTransform          (train + test branch)
  bilateral/lib_bilagrid.py:171-230,256-368  slice(), BilateralGrid.forward
  bilateral/lib_bilagrid.py:152-168          total_variation_loss
  models/trainers/scene_graph.py:95-98,112-117   how the trainer applies the maps
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/project"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


def import_reference():
    sys.dont_write_bytecode = True
    _stub("tensorly", set_backend=lambda *_: None)
    _stub("tensorly.decomposition", parafac=None)
    _stub("pytorch3d")
    _stub("pytorch3d.ops", knn_points=None)
    _stub("nvdiffrast")
    _stub("nvdiffrast.torch")
    sys.path.insert(0, REF)
    import models.modules as M  # noqa
    import bilateral.lib_bilagrid as LB  # noqa
    return M, LB


N_IMG = 3
K_IMG = 1
NEAR = [0, 2]


def make_inputs(H, W, grid_sizes, seed, wide, dtype):
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(H, W, 3, generator=g, dtype=torch.float64)
    if wide:
        rgb = rgb * 1.3 - 0.1  # exercises the border clamp of the guidance axis
    wt = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
    noise = [0.1 * torch.randn(N_IMG, 12, L, gy, gx, generator=g, dtype=torch.float64) for (gx, gy, L) in grid_sizes]
    return rgb.to(dtype), wt.to(dtype), [n.to(dtype) for n in noise]


def trainer_apply(affs, rgb):
    """scene_graph.py:112-117 (the trainer's own arithmetic, kept verbatim in spirit)."""
    out = rgb
    for aff in affs:
        aff = aff.reshape(rgb.shape[0], rgb.shape[1], 3, 4)
        out = (aff[..., :3, :3] @ out[..., None] + aff[..., :3, 3:])[..., 0]
    return out


def run_multiscale(M, H, W, grid_sizes, factors, seed, wide, dtype, tag, store_maps):
    torch.set_default_dtype(dtype)
    rgb, wt, noise = make_inputs(H, W, grid_sizes, seed, wide, dtype)
    aff = M.MultiScaleBilateralAffineTransform("Affine", n=N_IMG, grid=[list(s) for s in grid_sizes], device="cpu").to(dtype)
    with torch.no_grad():
        for i, n in enumerate(noise):
            getattr(aff, f"bil_grids{i}").grids.add_(n)
    infos = {"img_idx": torch.full((H, W), K_IMG, dtype=torch.long)}
    rgb_in = rgb.clone().requires_grad_(True)
    outs = aff(rgb_in, infos, guidance_factor=list(factors))
    out = trainer_apply(outs, rgb_in)
    tv = aff.tv_loss()
    loss = (out * wt).sum() + 0.37 * tv
    loss.backward()
    rec = {
        "H": H, "W": W, "grid_sizes": np.array(grid_sizes), "factors": np.array(factors), "k_img": K_IMG,
        "rgb": rgb.numpy(), "wt": wt.numpy(), "tv_coef": 0.37,
        "out": out.detach().numpy(), "tv": tv.detach().numpy(), "v_rgb": rgb_in.grad.numpy(),
    }
    for i in range(len(grid_sizes)):
        gp = getattr(aff, f"bil_grids{i}").grids
        rec[f"grids{i}"] = gp.detach().numpy()
        rec[f"v_grids{i}"] = gp.grad.numpy()
        m = outs[i].detach().reshape(H, W, 12)
        rec[f"aff{i}"] = (m if store_maps else m[::5, ::7]).numpy()
    # test branch (modules.py:523-547): neighbour-frame average
    aff.in_test_set = True
    aff.training_indices_for_test = {K_IMG: NEAR}
    with torch.no_grad():
        outs_t = aff(rgb, infos, guidance_factor=list(factors))
        rec["out_test"] = trainer_apply(outs_t, rgb).numpy()
    rec["near"] = np.array(NEAR)
    np.savez_compressed(os.path.join(OUT, f"bilagrid_ms_{tag}.npz"), **rec)
    torch.set_default_dtype(torch.float32)


def run_single(M, H, W, seed, wide, dtype, tag, gs=(7, 5, 4)):
    torch.set_default_dtype(dtype)
    rgb, wt, noise = make_inputs(H, W, [gs], seed, wide, dtype)
    aff = M.BilateralAffineTransform("Affine", n=N_IMG, grid_X=gs[0], grid_Y=gs[1], grid_W=gs[2], device="cpu").to(dtype)
    with torch.no_grad():
        aff.bil_grids.grids.add_(noise[0])
    infos = {"img_idx": torch.full((H, W), K_IMG, dtype=torch.long)}
    rgb_in = rgb.clone().requires_grad_(True)
    a = aff(rgb_in, infos)
    a = a.reshape(H, W, 3, 4)  # scene_graph.py:96
    out = (a[..., :3, :3] @ rgb_in[..., None] + a[..., :3, 3:])[..., 0]
    tv = aff.tv_loss()
    loss = (out * wt).sum() + 10.0 * tv
    loss.backward()
    rec = {
        "H": H, "W": W, "k_img": K_IMG, "rgb": rgb.numpy(), "wt": wt.numpy(), "tv_coef": 10.0,
        "grids0": aff.bil_grids.grids.detach().numpy(), "v_grids0": aff.bil_grids.grids.grad.numpy(),
        "aff0": a.detach().reshape(H, W, 12).numpy(), "out": out.detach().numpy(),
        "tv": tv.detach().numpy(), "v_rgb": rgb_in.grad.numpy(),
    }
    aff.in_test_set = True
    aff.training_indices_for_test = {K_IMG: NEAR}
    with torch.no_grad():
        at = aff(rgb, infos).reshape(H, W, 3, 4)
        rec["out_test"] = ((at[..., :3, :3] @ rgb[..., None] + at[..., :3, 3:])[..., 0]).numpy()
    rec["near"] = np.array(NEAR)
    np.savez_compressed(os.path.join(OUT, f"bilagrid_single_{tag}.npz"), **rec)
    torch.set_default_dtype(torch.float32)


def run_points(LB, seed, dtype, tag):
    """lib_bilagrid.slice on 2-D inputs with several grid indices (the 'random pixels from
    different views' branch, lib_bilagrid.py:214-223)."""
    torch.set_default_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    P = 257
    bg = LB.BilateralGrid(N_IMG, grid_X=5, grid_Y=7, grid_W=3).to(dtype)
    with torch.no_grad():
        bg.grids.add_(0.1 * torch.randn(bg.grids.shape, generator=g, dtype=torch.float64).to(dtype))
    xy = torch.rand(P, 2, generator=g, dtype=torch.float64).to(dtype)
    rgb = (torch.rand(P, 3, generator=g, dtype=torch.float64) * 1.3 - 0.1).to(dtype)
    idx = torch.randint(0, N_IMG, (P, 1), generator=g)
    wt = torch.randn(P, 3, generator=g, dtype=torch.float64).to(dtype)
    rgb_in = rgb.clone().requires_grad_(True)
    res = LB.slice(bg, xy, rgb_in, idx)
    (res["rgb"] * wt).sum().backward()
    np.savez_compressed(
        os.path.join(OUT, f"bilagrid_points_{tag}.npz"),
        grids=bg.grids.detach().numpy(), xy=xy.numpy(), rgb=rgb.numpy(), idx=idx.numpy(), wt=wt.numpy(),
        out=res["rgb"].detach().numpy(), aff=res["rgb_affine_mats"].detach().reshape(P, 12).numpy(),
        v_rgb=rgb_in.grad.numpy(), v_grids=bg.grids.grad.numpy(),
    )
    # TV on a few shapes (lib_bilagrid.py:152-168)
    tvs = {}
    for i, shp in enumerate([(5, 12, 1, 2, 2), (5, 12, 2, 4, 4), (5, 12, 4, 8, 8), (2, 12, 3, 5, 7)]):
        x = torch.randn(*shp, generator=g, dtype=torch.float64).to(dtype).requires_grad_(True)
        tv = LB.total_variation_loss(x)
        if tv.requires_grad:
            tv.backward()
            gx = x.grad.numpy()
        else:
            gx = np.zeros(shp)
        tvs[f"x{i}"] = x.detach().numpy()
        tvs[f"tv{i}"] = np.asarray(float(tv.detach()))
        tvs[f"v_x{i}"] = gx
    np.savez_compressed(os.path.join(OUT, f"bilagrid_tv_{tag}.npz"), **tvs)
    torch.set_default_dtype(torch.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    M, LB = import_reference()
    L3 = [(2, 2, 1), (4, 4, 2), (8, 8, 4)]          # shipped 3-level config (configs/omnire_ms_bilateral*.yaml)
    L4 = [(2, 2, 1), (4, 4, 2), (8, 8, 4), (6, 5, 3)]  # 4 levels need explicit factors
    L1 = [(7, 5, 4)]                                # one level, factor 1: no resampling at all
    f32, f64 = torch.float32, torch.float64
    seed = 0
    # small image: every configuration, both dtypes, full per-level maps stored
    for wide in (False, True):
        w = "wide" if wide else "unit"
        for dtype, dn in ((f32, "f32"), (f64, "f64")):
            run_multiscale(M, 17, 23, L3, [4, 4, 2], seed, wide, dtype, f"L3_17x23_{w}_{dn}", True)
            run_multiscale(M, 17, 23, L4, [8, 4, 4, 2], seed + 1, wide, dtype, f"L4_17x23_{w}_{dn}", True)
            run_multiscale(M, 17, 23, L1, [1], seed + 2, wide, dtype, f"L1_17x23_{w}_{dn}", True)
            run_single(M, 17, 23, seed + 3, wide, dtype, f"17x23_{w}_{dn}")
        seed += 4
    # larger, non-multiple-of-factor image
    run_multiscale(M, 57, 91, L3, [4, 4, 2], 20, False, f32, "L3_57x91_unit_f32", False)
    run_multiscale(M, 57, 91, L3, [4, 4, 2], 21, True, f32, "L3_57x91_wide_f32", False)
    run_multiscale(M, 57, 91, L4, [8, 4, 4, 2], 22, True, f32, "L4_57x91_wide_f32", False)
    run_single(M, 57, 91, 23, True, f32, "57x91_wide_f32")
    # multiple-of-factor image + the shipped single-scale 16x16x8 grid
    run_multiscale(M, 64, 64, L3, [4, 4, 2], 30, False, f32, "L3_64x64_unit_f32", False)
    run_multiscale(M, 64, 64, L3, [4, 4, 2], 31, True, f32, "L3_64x64_wide_f32", False)
    run_single(M, 32, 48, 32, True, f32, "32x48_wide_g16_f32", gs=(16, 16, 8))
    for dtype, dn in ((torch.float32, "f32"), (torch.float64, "f64")):
        run_points(LB, 100, dtype, dn)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden files:", len(os.listdir(OUT)), "bytes:", tot)


if __name__ == "__main__":
    main()
