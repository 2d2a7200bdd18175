"""``rasterization`` with the signature the reference imports from gsplat v1.3.0
(/root/reference/project/models/gaussians/basics.py:12; called at
/root/reference/project/models/trainers/base.py:393-408 and :811-826).

Pipeline (every stage a hand-written gfx950 kernel behind libbds.so):
    projection -> tile intersection + ordering -> alpha compositing -> expected-depth normalise
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from .gs_ops import (TILE_SIZE, fully_fused_projection, isect_tiles, rasterize_to_pixels, spherical_harmonics)


class _Meta(dict):
    """The ``meta`` dict of gsplat's rasterization().  ``isect_ids`` (the sorted 64-bit keys, which nothing on
    the reference's path reads) is materialised on first access instead of on every step."""

    def __getitem__(self, key):
        if key == "isect_ids" and dict.__getitem__(self, key) is None:
            self["isect_ids"] = _isect_ids_from(self["isect_offsets"], self["flatten_ids"], self["depths"])
        return dict.__getitem__(self, key)


@torch.no_grad()
def _isect_ids_from(isect_offsets: Tensor, flatten_ids: Tensor, depths: Tensor) -> Tensor:
    M = flatten_ids.numel()
    idx = torch.arange(M, device=flatten_ids.device)
    tile = torch.bucketize(idx, isect_offsets.reshape(-1).long(), right=True) - 1  # camera*tiles + tile
    bits = depths.detach().reshape(-1)[flatten_ids.long()].contiguous().view(torch.int32).long()
    return (tile << 32) | bits


# Exact tile culling (see gs_ops.isect_tiles): on by default -- images and gradients are identical, only the
# intersection lists in ``meta`` (tiles_per_gauss / flatten_ids / isect_ids / isect_offsets) are shorter than
# gsplat's bounding-square lists.  Switch off to reproduce gsplat's lists entry for entry.
_TILE_CULLING = os.environ.get("BDS_TILE_CULL", "1") != "0"


def set_tile_culling(on: bool) -> None:
    global _TILE_CULLING
    _TILE_CULLING = bool(on)


def _as_int(v) -> int:
    # the reference passes 0-d (GPU) int64 tensors for width/height
    # (/root/reference/project/datasets/base/pixel_source.py:653-654, tools/train.py:262-264)
    if torch.is_tensor(v):
        return int(v.item())
    return int(v)


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]
    scales: Tensor,  # [N, 3]
    opacities: Tensor,  # [N]
    colors: Tensor,  # [(C,) N, D] or [(C,) N, K, 3]
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width,
    height,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    covars: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Returns (render_colors [C,H,W,D], render_alphas [C,H,W,1], meta).

    ``packed`` / ``sparse_grad`` / ``channel_chunk`` only select memory layouts inside gsplat; results
    are identical, so they are accepted and ignored (the dense layout is used: one camera per step on
    the reference's path).  ``distributed`` (gsplat's Gaussian-sharded mode) and ``covars`` are not
    used by the reference and raise."""
    if distributed:
        raise NotImplementedError("distributed=True (Gaussian-sharded rendering) is not on the reference's path; "
                                  "multi-GPU here shards views, see bilateral_driving_amd.dist")
    if covars is not None:
        raise NotImplementedError("covars= is not on the reference's path (quats/scales are passed)")
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED"), render_mode
    assert rasterize_mode in ("classic", "antialiased"), rasterize_mode
    assert tile_size >= TILE_SIZE and tile_size % TILE_SIZE == 0, (
        f"tile_size={tile_size}: a multiple of {TILE_SIZE} is required (lists of larger tiles are filtered per 16-px compositing tile)")
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    width, height = _as_int(width), _as_int(height)

    if sh_degree is None:
        # post-activation colours [N, D] or [C, N, D]
        assert (colors.dim() == 2 and colors.shape[0] == N) or (colors.dim() == 3 and colors.shape[:2] == (C, N)), colors.shape
    else:
        assert (colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3) or (
            colors.dim() == 4 and colors.shape[:2] == (C, N) and colors.shape[3] == 3), colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[-2], colors.shape

    radii, means2d, depths, conics, compensations = fully_fused_projection(
        means, quats, scales, viewmats, Ks, width, height, eps2d=eps2d, near_plane=near_plane, far_plane=far_plane,
        radius_clip=radius_clip, calc_compensations=(rasterize_mode == "antialiased"))
    opac = opacities[None, :].expand(C, N)
    if compensations is not None:
        opac = opac * compensations

    if sh_degree is None:
        col = colors if colors.dim() == 3 else colors[None].expand(C, N, colors.shape[-1])
    else:
        camtoworlds = torch.linalg.inv(viewmats)
        dirs = means[None, :, :] - camtoworlds[:, None, :3, 3]  # [C, N, 3]
        shs = colors if colors.dim() == 4 else colors[None].expand(C, N, colors.shape[-2], 3)
        col = spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0)
        col = torch.clamp_min(col + 0.5, 0.0)

    if render_mode in ("RGB+D", "RGB+ED"):
        col = torch.cat([col, depths[..., None]], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=backgrounds.device)], dim=-1)
    elif render_mode in ("D", "ED"):
        col = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, device=backgrounds.device)

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    tiles_per_gauss, isect_ids, flatten_ids, isect_offsets = isect_tiles(means2d, radii, depths, tile_size, tile_width,
                                                                         tile_height, want_isect_ids=False,
                                                                         conics=conics if _TILE_CULLING else None,
                                                                         opacities=opac.contiguous() if _TILE_CULLING else None)
    render_colors, render_alphas = rasterize_to_pixels(means2d, conics, col.contiguous(), opac.contiguous(), width, height,
                                                       tile_size, isect_offsets, flatten_ids, backgrounds=backgrounds,
                                                       absgrad=absgrad)
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat(
            [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)

    meta = _Meta({
        "camera_ids": None,
        "gaussian_ids": None,
        "radii": radii,
        "means2d": means2d,
        "depths": depths,
        "conics": conics,
        "opacities": opac,
        "tile_width": tile_width,
        "tile_height": tile_height,
        "tiles_per_gauss": tiles_per_gauss,
        "isect_ids": isect_ids,
        "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets,
        "width": width,
        "height": height,
        "tile_size": tile_size,
        "n_cameras": C,
    })
    return render_colors, render_alphas, meta
