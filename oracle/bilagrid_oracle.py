"""Oracle (TEST INFRASTRUCTURE, not product code): closed-form CPU restatement of the
bilateral-grid half of the hot path, written with elementary differentiable torch ops
(index / multiply / add) so that autograd of this file is the gradient reference.

PINNED against the reference: ``tests/golden/bilagrid_*.npz`` are produced by importing
the reference's own Python in the build container (``oracle/gen_golden_bilateral.py``)
and ``tests/test_oracle_bilagrid_golden.py`` checks every function below against them.

Reference being restated (file:line under /root/reference/project):
  * bilateral/lib_bilagrid.py:256-368   BilateralGrid.forward  (grid_sample, trilinear,
                                        align_corners=True, padding_mode="border")
  * bilateral/lib_bilagrid.py:171-230   slice()
  * bilateral/lib_bilagrid.py:152-168   total_variation_loss()
  * models/modules.py:494-504           get_sample_grid()   (bilinear down-sample + linspace xy)
  * models/modules.py:409-420           fill_matrix_res()   (bilinear up-sample of the 12 channels)
  * models/modules.py:505-547           MultiScaleBilateralAffineTransform.forward (train / test branch)
  * models/modules.py:317-346           BilateralAffineTransform.forward (single scale)
  * models/trainers/scene_graph.py:95-98,112-117   application / sequential composition
"""
from __future__ import annotations

import torch

GRAY_W = (0.299, 0.587, 0.114)  # lib_bilagrid.py:287 (BT.601)


# --------------------------------------------------------------------------------------
# resampling  (torch F.interpolate(mode="bilinear", align_corners=False, antialias=False))
# --------------------------------------------------------------------------------------
def _src_index(out_size: int, in_size: int, dtype, device):
    """Source index/weight of the bilinear resampler (modules.py:497 and :414-419)."""
    if out_size == in_size:
        i0 = torch.arange(out_size, device=device)
        return i0, i0, torch.zeros(out_size, dtype=dtype, device=device)
    scale = torch.tensor(in_size, dtype=dtype, device=device) / out_size
    dst = torch.arange(out_size, dtype=dtype, device=device)
    src = torch.clamp(scale * (dst + 0.5) - 0.5, min=0.0)
    i0 = src.floor().long()
    i1 = i0 + (i0 < in_size - 1).long()
    w1 = torch.clamp(src - i0.to(dtype), 0.0, 1.0)
    return i0, i1, w1


def resize_bilinear(img: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """img [H, W, C] -> [Ho, Wo, C]."""
    H, W, _ = img.shape
    if H == Ho and W == Wo:
        return img
    y0, y1, wy = _src_index(Ho, H, img.dtype, img.device)
    x0, x1, wx = _src_index(Wo, W, img.dtype, img.device)
    wy = wy[:, None, None]
    wx = wx[None, :, None]
    top = img[y0][:, x0] * (1 - wx) + img[y0][:, x1] * wx
    bot = img[y1][:, x0] * (1 - wx) + img[y1][:, x1] * wx
    return top * (1 - wy) + bot * wy


def linspace01(n: int, dtype, device) -> torch.Tensor:
    """torch.linspace(0, 1, n) (modules.py:498-503) -- kept as the torch call so that the
    float32 rounding is the reference's."""
    return torch.linspace(0, 1.0, n, dtype=dtype, device=device)


# --------------------------------------------------------------------------------------
# slice  (lib_bilagrid.py:346-363)
# --------------------------------------------------------------------------------------
def rgb2gray(rgb: torch.Tensor) -> torch.Tensor:
    return rgb[..., 0] * GRAY_W[0] + rgb[..., 1] * GRAY_W[1] + rgb[..., 2] * GRAY_W[2]


def _unnormalize(c01: torch.Tensor, size: int) -> torch.Tensor:
    """[0,1] coordinate -> clipped grid index, in the reference's op order:
    (c-0.5)*2 (lib_bilagrid.py:352), then grid_sample's align_corners=True un-normalise
    ((c+1)/2)*(size-1) and the padding_mode="border" clip."""
    t = (c01 - 0.5) * 2
    return torch.clamp(((t + 1) / 2) * (size - 1), 0, size - 1)


def slice_grid(grid: torch.Tensor, x: torch.Tensor, y: torch.Tensor, gray: torch.Tensor) -> torch.Tensor:
    """grid [12, L, gy, gx]; x, y, gray [...] (x, y in [0,1]; gray any) -> affine [..., 12].

    Guidance follows lib_bilagrid.py:288,353: z_norm = gray*2 - 1."""
    C, L, gy, gx = grid.shape
    ix = _unnormalize(x, gx)
    iy = _unnormalize(y, gy)
    vz = (((gray * 2 - 1) + 1) / 2) * (L - 1)
    # grid_sample's border clip passes NO gradient on the closed boundary (v <= 0 or v >= L-1),
    # unlike torch.clamp; x/y carry no gradient in this path.
    iz = torch.where((vz > 0) & (vz < L - 1), vz, vz.detach().clamp(0, L - 1))
    x0f, y0f, z0f = ix.floor(), iy.floor(), iz.floor()
    fx, fy, fz = ix - x0f, iy - y0f, iz - z0f
    x0, y0, z0 = x0f.long(), y0f.long(), z0f.long()
    x1 = torch.clamp(x0 + 1, max=gx - 1)
    y1 = torch.clamp(y0 + 1, max=gy - 1)
    z1 = torch.clamp(z0 + 1, max=L - 1)
    g = grid.permute(1, 2, 3, 0)  # [L, gy, gx, 12]
    out = 0
    for zz, wz in ((z0, 1 - fz), (z1, fz)):
        for yy, wy in ((y0, 1 - fy), (y1, fy)):
            for xx, wx in ((x0, 1 - fx), (x1, fx)):
                out = out + g[zz, yy, xx] * (wz * wy * wx)[..., None]
    return out


def apply_affine(aff12: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
    """aff12 [..., 12] (row-major 3x4), rgb [..., 3] -> A[:, :3] @ rgb + A[:, 3]
    (lib_bilagrid.py:131-145, scene_graph.py:97,117)."""
    A = aff12.reshape(*aff12.shape[:-1], 3, 4)
    return (A[..., :3] * rgb[..., None, :]).sum(-1) + A[..., 3]


# --------------------------------------------------------------------------------------
# transforms (modules.py)
# --------------------------------------------------------------------------------------
def single_scale_affine(grid: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
    """BilateralAffineTransform.forward train branch (modules.py:317-335): rgb [H,W,3] ->
    affine map [H,W,12]."""
    H, W, _ = rgb.shape
    x = linspace01(W, rgb.dtype, rgb.device)[None, :].expand(H, W)
    y = linspace01(H, rgb.dtype, rgb.device)[:, None].expand(H, W)
    return slice_grid(grid, x, y, rgb2gray(rgb))


def lowres_affine(grid: torch.Tensor, rgb: torch.Tensor, factor: int) -> torch.Tensor:
    """get_sample_grid + slice at low resolution (modules.py:494-520) -> [Hd, Wd, 12]."""
    H, W, _ = rgb.shape
    Hd, Wd = H // factor, W // factor
    lo = resize_bilinear(rgb, Hd, Wd)
    x = linspace01(Wd, rgb.dtype, rgb.device)[None, :].expand(Hd, Wd)
    y = linspace01(Hd, rgb.dtype, rgb.device)[:, None].expand(Hd, Wd)
    return slice_grid(grid, x, y, rgb2gray(lo))


def multiscale_affines(grids, rgb: torch.Tensor, factors, neighbours=None):
    """MultiScaleBilateralAffineTransform.forward (modules.py:505-547).

    grids:   list over levels of [12, L, gy, gx]  (train branch: the current image's grids), or
             list over levels of [K, 12, L, gy, gx] with ``neighbours=True`` (test branch: the K
             neighbouring training images' grids, whose low-res slices are averaged, :523-535).
    returns: list of full-resolution affine maps [H, W, 12].
    """
    H, W, _ = rgb.shape
    outs = []
    for g, f in zip(grids, factors):
        if neighbours:
            lo = 0
            for k in range(g.shape[0]):
                lo = lo + lowres_affine(g[k], rgb, f)
            lo = lo / g.shape[0]
        else:
            lo = lowres_affine(g, rgb, f)
        outs.append(resize_bilinear(lo, H, W))  # fill_matrix_res
    return outs


def compose(affs, rgb: torch.Tensor) -> torch.Tensor:
    """Sequential composition (scene_graph.py:112-117): every level's map was sliced from the
    ORIGINAL rgb; they are applied one after the other."""
    out = rgb
    for a in affs:
        out = apply_affine(a, out)
    return out


def multiscale_transform(grids, rgb, factors, neighbours=None):
    return compose(multiscale_affines(grids, rgb, factors, neighbours), rgb)


def sky_blend(rgb_g: torch.Tensor, alpha: torch.Tensor, sky: torch.Tensor) -> torch.Tensor:
    """clamp (trainers/base.py:417) + sky blend (scene_graph.py:292-294)."""
    return torch.clamp(rgb_g, max=1.0) + sky * (1 - alpha)


# --------------------------------------------------------------------------------------
# TV regulariser (lib_bilagrid.py:152-168, modules.py:445,466-472)
# --------------------------------------------------------------------------------------
def total_variation_loss(x: torch.Tensor) -> torch.Tensor:
    """x [N, 12, L, gy, gx]."""
    tv = 0
    for d in range(2, x.dim()):
        n = x.shape[d]
        if n < 2:
            continue  # empty difference: sum()=0, count clamped to 1
        a = x.narrow(d, 1, n - 1)
        b = x.narrow(d, 0, n - 1)
        count = max(float(a[0].numel()), 1.0)
        tv = tv + ((a - b) ** 2).sum() / count
    return tv / x.shape[0]


def tv_weight(gx: int, gy: int, L: int) -> float:
    return 0.5 * (gx * gy * L) ** 0.5  # modules.py:445


def multiscale_tv(all_grids) -> torch.Tensor:
    """all_grids: list over levels of [N_img, 12, L, gy, gx]."""
    loss = 0
    for g in all_grids:
        _, _, L, gy, gx = g.shape
        loss = loss + total_variation_loss(g) * tv_weight(gx, gy, L)
    return loss
