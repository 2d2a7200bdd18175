"""Bilateral-grid half of the hot path, bound to libbds.so.

Mirrors the reference's ``bilateral.lib_bilagrid`` surface (same class / function names, argument
meaning, parameter name ``grids`` and layout [num, 12, L, H, W], error behaviour) so that
/root/reference/project/models/modules.py:13 resolves unmodified against
``bilateral_driving_amd/dropin``:

    BilateralGrid, slice, total_variation_loss, color_affine_transform
    (/root/reference/project/bilateral/lib_bilagrid.py:131-145,152-168,171-230,256-368)

plus the fused image transform the reference spreads over ~30 torch launches per level
(models/modules.py:494-522,409-420 + trainers/scene_graph.py:112-117):

    bilagrid_transform(rgb, grids, factors, ...) -> rgb_out [, per-level affine maps]
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib as L


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def color_affine_transform(affine_mats: Tensor, rgb: Tensor) -> Tensor:
    """affine_mats [..., 3, 4], rgb [..., 3] -> [..., 3]."""
    return torch.matmul(affine_mats[..., :3], rgb.unsqueeze(-1)).squeeze(-1) + affine_mats[..., 3]


# --------------------------------------------------------------------------------------------
# TV regulariser
# --------------------------------------------------------------------------------------------
class _TotalVariation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: float):
        L.require_gpu(x)
        x = _f32c(x)
        n, c, gl, gy, gx = x.shape
        out = torch.zeros(1, device=x.device, dtype=torch.float32)
        L.check(L.lib().bds_grid_tv_fwd(n, c, gx, gy, gl, L.ptr(x), weight, L.ptr(out), L.stream()), "bds_grid_tv_fwd")
        ctx.save_for_backward(x)
        ctx.weight = weight
        return out.reshape(())

    @staticmethod
    def backward(ctx, v_out):
        (x,) = ctx.saved_tensors
        n, c, gl, gy, gx = x.shape
        v = _f32c(v_out.reshape(1))
        v_x = torch.zeros_like(x)
        L.check(L.lib().bds_grid_tv_bwd(n, c, gx, gy, gl, L.ptr(x), ctx.weight, L.ptr(v), L.ptr(v_x), L.stream()),
                "bds_grid_tv_bwd")
        return v_x, None


def total_variation_loss(x: Tensor, weight: float = 1.0) -> Tensor:
    """Total variation of bilateral grids x [B, C, L, H, W] (lib_bilagrid.py:152-168; C = 12 affine or any feature
    width), times ``weight``."""
    if x.dim() != 5:
        raise ValueError("total_variation_loss expects bilateral grids of shape (B, C, L, H, W)")
    return _TotalVariation.apply(x, float(weight))


# --------------------------------------------------------------------------------------------
# point slice (BilateralGrid.forward)
# --------------------------------------------------------------------------------------------
class _SlicePoints(torch.autograd.Function):
    """grid [C,L,gy,gx], xy [P,2], rgb [P,3] -> [P,C]  (C = 12: affine maps; any other C: feature grids)"""

    @staticmethod
    def forward(ctx, grid: Tensor, xy: Tensor, rgb: Tensor):
        L.require_gpu(grid, xy, rgb)
        grid, xy, rgb = _f32c(grid), _f32c(xy), _f32c(rgb)
        P = xy.shape[0]
        nc, gl, gy, gx = grid.shape
        aff = torch.empty(P, nc, device=grid.device, dtype=torch.float32)
        if nc == 12:
            L.check(L.lib().bds_bilagrid_slice_fwd(P, L.ptr(grid), gx, gy, gl, L.ptr(xy), L.ptr(rgb), L.ptr(aff), L.stream()),
                    "bds_bilagrid_slice_fwd")
        else:
            L.check(L.lib().bds_bilagrid_slice_feat_fwd(P, nc, L.ptr(grid), gx, gy, gl, L.ptr(xy), L.ptr(rgb), L.ptr(aff), L.stream()),
                    "bds_bilagrid_slice_feat_fwd")
        ctx.save_for_backward(grid, xy, rgb)
        return aff

    @staticmethod
    def backward(ctx, v_aff):
        grid, xy, rgb = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. the slice xy coordinates is not on the reference's path "
                                      "(they come from linspace / pixel indices)")
        P = xy.shape[0]
        nc, gl, gy, gx = grid.shape
        v_aff = _f32c(v_aff)
        v_grid = torch.zeros_like(grid) if ctx.needs_input_grad[0] else None
        v_rgb = torch.empty_like(rgb) if ctx.needs_input_grad[2] else None
        if nc == 12:
            L.check(L.lib().bds_bilagrid_slice_bwd(P, L.ptr(grid), gx, gy, gl, L.ptr(xy), L.ptr(rgb), L.ptr(v_aff), L.ptr(v_grid),
                                                   L.ptr(v_rgb), L.stream()), "bds_bilagrid_slice_bwd")
        else:
            L.check(L.lib().bds_bilagrid_slice_feat_bwd(P, nc, L.ptr(grid), gx, gy, gl, L.ptr(xy), L.ptr(rgb), L.ptr(v_aff),
                                                        L.ptr(v_grid), L.ptr(v_rgb), L.stream()), "bds_bilagrid_slice_feat_bwd")
        return v_grid, None, v_rgb


class _SliceImage(torch.autograd.Function):
    """grid [C,L,gy,gx], rgb [H,W,3] -> [H,W,C]: the feature slice at the pixel grid (xy = linspace x linspace), band-staged in LDS."""

    @staticmethod
    def forward(ctx, grid: Tensor, rgb: Tensor):
        L.require_gpu(grid, rgb)
        grid, rgb = _f32c(grid), _f32c(rgb)
        H, W, _ = rgb.shape
        nc, gl, gy, gx = grid.shape
        out = torch.empty(H, W, nc, device=grid.device, dtype=torch.float32)
        L.check(L.lib().bds_bilagrid_slice_feat_image_fwd(H, W, nc, L.ptr(grid), gx, gy, gl, L.ptr(rgb), L.ptr(out), L.stream()),
                "bds_bilagrid_slice_feat_image_fwd")
        ctx.save_for_backward(grid, rgb)
        return out

    @staticmethod
    def backward(ctx, v_out):
        grid, rgb = ctx.saved_tensors
        H, W, _ = rgb.shape
        nc, gl, gy, gx = grid.shape
        v_out = _f32c(v_out)
        v_grid = torch.zeros_like(grid) if ctx.needs_input_grad[0] else None
        v_rgb = torch.empty_like(rgb) if ctx.needs_input_grad[1] else None
        L.check(L.lib().bds_bilagrid_slice_feat_image_bwd(H, W, nc, L.ptr(grid), gx, gy, gl, L.ptr(rgb), L.ptr(v_out), L.ptr(v_grid),
                                                          L.ptr(v_rgb), L.stream()), "bds_bilagrid_slice_feat_image_bwd")
        return v_grid, v_rgb


def slice_feature_image(bil_grids: "NeuralBilateralGrid", rgb: Tensor, idx: int) -> Tensor:
    """``slice_feature(bil_grids, xy, rgb, idx)["affine_features"]`` for the case every caller in the reference has
    (models/modules.py:643-650, 728-760): ``xy`` = the pixel grid ``meshgrid(linspace(0,1,H), linspace(0,1,W))`` of the image ``rgb``
    [H,W,3].  Returns [1,H,W,feature_dim].  The coordinates are implied, so the kernels stage the grid rows of a pixel row in LDS
    (``bds_bilagrid_slice_feat_image_*``); grids whose row band does not fit go through the point form."""
    grid = bil_grids.grids[int(idx)]
    nc, gl, gy, gx = grid.shape
    H, W, _ = rgb.shape
    if rgb.is_cuda and L.lib().bds_bilagrid_slice_feat_image_ok(nc, gx, gy, gl):
        return _SliceImage.apply(grid, rgb).unsqueeze(0)
    ys, xs = torch.meshgrid(torch.linspace(0, 1.0, H, device=rgb.device), torch.linspace(0, 1.0, W, device=rgb.device), indexing="ij")
    xy = torch.stack([xs, ys], dim=-1).unsqueeze(0)
    return slice_feature(bil_grids, xy, rgb.unsqueeze(0), torch.tensor(int(idx), device=rgb.device, dtype=torch.long))["affine_features"]


class BilateralGrid(nn.Module):
    """Holds ``num`` bilateral grids [num, 12, L, H, W], identity-initialised (lib_bilagrid.py:256-311)."""

    def __init__(self, num, grid_X=16, grid_Y=16, grid_W=8, mode="bilinear"):
        super().__init__()
        if mode != "bilinear":
            raise NotImplementedError("only mode='bilinear' (the reference's only value, modules.py:434-444)")
        self.grid_width = grid_X
        self.grid_height = grid_Y
        self.grid_guidance = grid_W
        self.mode = mode
        ident = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0], dtype=torch.float32)
        grid = ident.reshape(1, 12, 1, 1, 1).repeat(num, 1, grid_W, grid_Y, grid_X)
        self.grids = nn.Parameter(grid.contiguous())
        self.register_buffer("rgb2gray_weight", torch.tensor([[0.299, 0.587, 0.114]], dtype=torch.float32))

    def tv_loss(self):
        return total_variation_loss(self.grids)

    def forward(self, grid_xy: Tensor, rgb: Tensor, idx: Optional[Tensor] = None) -> Tensor:
        """grid_xy [..., 2] in [0,1], rgb [..., 3]; 2-D..4-D inputs need ``idx`` ([B] grid indices, one per
        leading-batch entry); 5-D inputs use one grid per leading entry.  Returns [..., 3, 4]."""
        out = _slice_grids(self.grids, grid_xy, rgb, idx)
        return out.reshape(*grid_xy.shape[:-1], 3, 4)


def _slice_grids(grids: Tensor, grid_xy: Tensor, rgb: Tensor, idx: Optional[Tensor]) -> Tensor:
    """Shared body of BilateralGrid.forward / NeuralBilateralGrid.forward: [B, per, C] sliced values."""
    nd = grid_xy.dim()
    assert rgb.dim() == nd
    if not (1 < nd <= 5):
        raise ValueError("Bilateral grid slicing only takes either 2D, 3D, 4D and 5D inputs")
    if nd < 5:
        assert idx is not None
    B, nc = grid_xy.shape[0], grids.shape[1]
    if idx is None:
        assert grids.shape[0] == B
        sel = range(B)
    else:
        idx = idx.reshape(-1)
        assert idx.numel() == B, (idx.shape, B)
        sel = idx.tolist()
    per = grid_xy[0].numel() // 2
    flat_xy = grid_xy.reshape(B, per, 2)
    flat_rgb = rgb.reshape(B, per, 3)
    # one launch per distinct grid: batch entries using the same grid are sliced together
    groups = {}
    for b, g in enumerate(sel):
        groups.setdefault(int(g), []).append(b)
    if len(groups) == 1:
        (g, _), = groups.items()
        return _SlicePoints.apply(grids[g], flat_xy.reshape(-1, 2), flat_rgb.reshape(-1, 3)).reshape(B, per, nc)
    out = torch.zeros(B, per, nc, device=rgb.device, dtype=torch.float32)
    for g, bs in groups.items():
        bsel = torch.tensor(bs, device=rgb.device)
        a = _SlicePoints.apply(grids[g], flat_xy[bsel].reshape(-1, 2), flat_rgb[bsel].reshape(-1, 3))
        out = out.index_add(0, bsel, a.reshape(len(bs), per, nc))
    return out


class NeuralBilateralGrid(nn.Module):
    """``num`` FEATURE grids [num, feature_dim, L, H, W], zero-initialised (lib_bilagrid.py:370-414); ``forward`` slices
    them like ``BilateralGrid`` and returns the features [..., feature_dim] (lib_bilagrid.py:420-461)."""

    def __init__(self, num, grid_X=16, grid_Y=16, grid_W=8, feature_dim=8, mode="bilinear"):
        super().__init__()
        if mode != "bilinear":
            raise NotImplementedError("only mode='bilinear'")
        self.grid_width, self.grid_height, self.grid_guidance, self.mode = grid_X, grid_Y, grid_W, mode
        self.grids = nn.Parameter(torch.zeros(num, feature_dim, grid_W, grid_Y, grid_X, dtype=torch.float32))
        self.register_buffer("rgb2gray_weight", torch.tensor([[0.299, 0.587, 0.114]], dtype=torch.float32))

    def tv_loss(self):
        return total_variation_loss(self.grids)

    def forward(self, grid_xy: Tensor, rgb: Tensor, idx: Optional[Tensor] = None) -> Tensor:
        out = _slice_grids(self.grids, grid_xy, rgb, idx)     # [B, per, f]
        nd = grid_xy.dim()
        # the reference pads the input to 5-D and squeezes dim 1 of the result (lib_bilagrid.py:437-459)
        shape5 = list(grid_xy.shape[:1]) + [1] * (5 - nd) + list(grid_xy.shape[1:-1]) + [out.shape[-1]]
        out = out.reshape(shape5)
        return out.squeeze(1)


def slice_feature(bil_grids: NeuralBilateralGrid, xy: Tensor, rgb: Tensor, grid_idx: Tensor):
    """lib_bilagrid.slice_feature (:232-253): as ``slice`` but returns ``{"affine_features": [..., feature_dim]}``."""
    shape = rgb.shape
    distinct = torch.unique(grid_idx)
    if distinct.numel() == 1:
        per_entry_idx = distinct
        xy, rgb = xy[None], rgb[None]
    else:
        if not 2 <= grid_idx.dim() <= 4:
            raise ValueError("The input to bilateral grid slicing is not supported yet.")
        per_entry_idx = grid_idx.reshape(grid_idx.shape[0], -1)[:, 0]
    feats = bil_grids(xy, rgb, per_entry_idx)
    return {"affine_features": feats.reshape(*shape[:-1], feats.shape[-1])}


def slice(bil_grids: BilateralGrid, xy: Tensor, rgb: Tensor, grid_idx: Tensor):
    """Slice bilateral grids at pixel coordinates ``xy`` (in [0,1]) guided by the gray value of ``rgb`` and apply the
    sliced 3x4 affine maps: returns ``{"rgb": transformed colours, "rgb_affine_mats": [..., 3, 4]}``
    (same contract as the reference's lib_bilagrid.slice, /root/reference/project/bilateral/lib_bilagrid.py:171-230).

    Batched inputs (2-D .. 4-D, leading dim = batch) carry one grid index per batch entry in ``grid_idx[:, 0, ...]``;
    when every sample uses the same grid (the training path: one image per step) the whole input is one batch entry."""
    shape = rgb.shape
    distinct = torch.unique(grid_idx)
    if distinct.numel() == 1:
        per_entry_idx = distinct
        xy, rgb = xy[None], rgb[None]
    else:
        if not 2 <= grid_idx.dim() <= 4:
            raise ValueError("The input to bilateral grid slicing is not supported yet.")
        per_entry_idx = grid_idx.reshape(grid_idx.shape[0], -1)[:, 0]
    mats = bil_grids(xy, rgb, per_entry_idx)
    out = color_affine_transform(mats, rgb)
    return {"rgb": out.reshape(shape), "rgb_affine_mats": mats.reshape(*shape[:-1], 3, 4)}


# --------------------------------------------------------------------------------------------
# fused image transform
# --------------------------------------------------------------------------------------------
def _levels_struct(grids: Sequence[Tensor], v_grids, factors: Sequence[int]):
    n = len(grids)
    arr = (L.BdsLevel * n)()
    for i, (g, f) in enumerate(zip(grids, factors)):
        n_avg, c, gl, gy, gx = g.shape
        assert c == 12
        arr[i].grid = g.data_ptr()
        arr[i].v_grid = v_grids[i].data_ptr() if v_grids is not None and v_grids[i] is not None else None
        arr[i].gx, arr[i].gy, arr[i].gl, arr[i].factor, arr[i].n_avg = gx, gy, gl, int(f), n_avg
    return arr


class _BilagridTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, alpha, sky, factors, want_maps, *grids):
        L.require_gpu(rgb, alpha, sky, *grids)
        rgb, alpha, sky = _f32c(rgb), _f32c(alpha), _f32c(sky)
        grids = [_f32c(g) for g in grids]
        H, W, _ = rgb.shape
        n = len(grids)
        lib = L.lib()
        lv = _levels_struct(grids, None, factors)
        ws_bytes = lib.bds_bilagrid_ms_workspace_bytes(n, lv, H, W)
        if ws_bytes == 0:
            raise L.BdsError("bds_bilagrid_ms_workspace_bytes rejected the level configuration")
        ws = torch.empty(ws_bytes, device=rgb.device, dtype=torch.uint8)
        out = torch.empty(H, W, 3, device=rgb.device, dtype=torch.float32)
        maps = [torch.empty(H, W, 12, device=rgb.device, dtype=torch.float32) for _ in range(n)] if want_maps else []
        maps_arr = (C.c_void_p * n)(*[m.data_ptr() for m in maps]) if want_maps else None
        with L.timed("bilagrid_fwd"):
            L.check(lib.bds_bilagrid_ms_fwd(n, lv, H, W, L.ptr(rgb), L.ptr(alpha), L.ptr(sky), L.ptr(ws), ws_bytes, L.ptr(out),
                                            maps_arr, L.stream()), "bds_bilagrid_ms_fwd")
        ctx.save_for_backward(rgb, alpha, sky, ws, *grids)
        ctx.factors = tuple(int(f) for f in factors)
        ctx.n = n
        ctx.mark_non_differentiable(*maps)
        return (out, *maps)

    @staticmethod
    def backward(ctx, v_out, *v_maps):
        rgb, alpha, sky, ws, *grids = ctx.saved_tensors
        H, W, _ = rgb.shape
        n = ctx.n
        lib = L.lib()
        need_g = ctx.needs_input_grad[5:]
        v_grids = [torch.zeros_like(g) if need_g[i] else None for i, g in enumerate(grids)]
        lv = _levels_struct(grids, v_grids, ctx.factors)
        v_out = _f32c(v_out)
        v_rgb = torch.empty_like(rgb)
        v_alpha = torch.empty_like(alpha) if sky is not None else None
        v_sky = torch.empty_like(sky) if sky is not None else None
        with L.timed("bilagrid_bwd"):
            L.check(lib.bds_bilagrid_ms_bwd(n, lv, H, W, L.ptr(rgb), L.ptr(alpha), L.ptr(sky), L.ptr(ws), ws.numel(),
                                            L.ptr(v_out), L.ptr(v_rgb), L.ptr(v_alpha), L.ptr(v_sky), L.stream()),
                    "bds_bilagrid_ms_bwd")
        return (v_rgb, v_alpha, v_sky, None, None, *v_grids)


def bilagrid_transform(rgb: Tensor, grids: Sequence[Tensor], factors: Sequence[int], alpha: Optional[Tensor] = None,
                       sky: Optional[Tensor] = None, return_maps: bool = False):
    """Fused multi-scale bilateral slice + 3x4 affine on an image.

    rgb [H,W,3]; grids: per level [12,L,gy,gx] (one image's grid) or [K,12,L,gy,gx] (K grids whose
    low-res slices are averaged: the test branch, modules.py:523-535); factors: per-level guidance
    down-sampling factor (modules.py:505 default [4,4,2]; 1 = single-scale transform, :317-335).
    If ``sky`` is given the input colour is clamp(rgb, max=1) + sky*(1-alpha)
    (trainers/base.py:417 + scene_graph.py:292-294) with alpha [H,W] or [H,W,1].

    Returns rgb_out [H,W,3], or (rgb_out, [per-level affine maps [H,W,3,4]]) with ``return_maps``."""
    assert rgb.dim() == 3 and rgb.shape[-1] == 3, rgb.shape
    assert len(grids) == len(factors) and len(grids) >= 1
    gs = [g if g.dim() == 5 else g[None] for g in grids]
    if sky is not None:
        assert alpha is not None
        alpha = alpha.reshape(rgb.shape[0], rgb.shape[1])
    else:
        alpha = None
    res = _BilagridTransform.apply(rgb, alpha, sky, tuple(factors), bool(return_maps), *gs)
    if return_maps:
        H, W, _ = rgb.shape
        return res[0], [m.reshape(H, W, 3, 4) for m in res[1:]]
    return res[0]
