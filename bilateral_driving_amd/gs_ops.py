"""Autograd operators of the rasterizer half, bound to libbds.so.

They mirror the gsplat v1.3.0 operators the reference reaches (names, argument meaning, output
shapes) so that ``rendering.rasterization`` below them reads like the reference's dependency:

  spherical_harmonics      <- gsplat.cuda._wrapper.spherical_harmonics
                              (/root/reference/project/models/gaussians/vanilla.py:388)
  fully_fused_projection,
  isect_tiles,
  rasterize_to_pixels      <- stages of gsplat.rendering.rasterization
                              (/root/reference/project/models/trainers/base.py:393-408)
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L

TILE_SIZE = 16  # the compositing tile of the gfx950 kernels (gsplat default); lists may be built for any multiple of it


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# --------------------------------------------------------------------------------------------
# spherical harmonics
# --------------------------------------------------------------------------------------------
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor]):
        L.require_gpu(dirs, coeffs, masks)
        dirs, coeffs = _f32c(dirs), _f32c(coeffs)
        n, K = coeffs.shape[0], coeffs.shape[1]
        m8 = None if masks is None else masks.to(torch.uint8).contiguous()
        out = torch.empty(n, 3, device=coeffs.device, dtype=torch.float32)
        with L.timed("sh_fwd"):
            L.check(L.lib().bds_sh_fwd(n, K, degree, L.ptr(dirs), L.ptr(coeffs), L.ptr(m8), L.ptr(out), L.stream()), "bds_sh_fwd")
        ctx.save_for_backward(dirs, coeffs, m8)
        ctx.degree = degree
        return out

    @staticmethod
    def backward(ctx, v_out: Tensor):
        dirs, coeffs, m8 = ctx.saved_tensors
        n, K = coeffs.shape[0], coeffs.shape[1]
        v_out = _f32c(v_out)
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if ctx.needs_input_grad[1] else None
        with L.timed("sh_bwd"):
            L.check(L.lib().bds_sh_bwd(n, K, ctx.degree, L.ptr(dirs), L.ptr(coeffs), L.ptr(m8), L.ptr(v_out), L.ptr(v_coeffs),
                                       L.ptr(v_dirs), L.stream()), "bds_sh_bwd")
        return None, v_dirs, v_coeffs if ctx.needs_input_grad[2] else None, None


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """dirs [..., 3], coeffs [..., K, 3], masks [...] -> colours [..., 3]."""
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3 and coeffs.shape[-1] == 3
    batch = dirs.shape[:-1]
    if masks is not None:
        assert masks.shape == batch, masks.shape
        masks = masks.reshape(-1)
    out = _SphericalHarmonics.apply(degrees_to_use, dirs.reshape(-1, 3), coeffs.reshape(-1, coeffs.shape[-2], 3), masks)
    return out.reshape(*batch, 3)


# --------------------------------------------------------------------------------------------
# projection
# --------------------------------------------------------------------------------------------
class _Projection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
                calc_compensations):
        L.require_gpu(means, quats, scales, viewmats, Ks)
        means, quats, scales, viewmats, Ks = map(_f32c, (means, quats, scales, viewmats, Ks))
        Cn, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        radii = torch.empty(Cn, N, device=dev, dtype=torch.int32)
        means2d = torch.empty(Cn, N, 2, device=dev, dtype=torch.float32)
        depths = torch.empty(Cn, N, device=dev, dtype=torch.float32)
        conics = torch.empty(Cn, N, 3, device=dev, dtype=torch.float32)
        comps = torch.empty(Cn, N, device=dev, dtype=torch.float32) if calc_compensations else None
        with L.timed("project_fwd"):
            L.check(L.lib().bds_project_fwd(Cn, N, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(viewmats), L.ptr(Ks), width,
                                            height, eps2d, near_plane, far_plane, radius_clip, L.ptr(radii), L.ptr(means2d),
                                            L.ptr(depths), L.ptr(conics), L.ptr(comps), L.stream()), "bds_project_fwd")
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, conics)
        ctx.cfg = (width, height, eps2d)
        ctx.mark_non_differentiable(radii)
        if comps is None:
            return radii, means2d, depths, conics
        return radii, means2d, depths, conics, comps

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_comps=None):
        means, quats, scales, viewmats, Ks, radii, conics = ctx.saved_tensors
        if v_comps is not None and bool((v_comps != 0).any()):
            raise NotImplementedError("gradient through compensations (rasterize_mode='antialiased') is not on the "
                                      "reference's path (render.antialiased is false in every shipped config)")
        width, height, eps2d = ctx.cfg
        Cn, N = viewmats.shape[0], means.shape[0]
        v_means2d, v_depths, v_conics = _f32c(v_means2d), _f32c(v_depths), _f32c(v_conics)
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        v_viewmats = torch.empty_like(viewmats) if ctx.needs_input_grad[3] else None
        with L.timed("project_bwd"):
            L.check(L.lib().bds_project_bwd(Cn, N, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(viewmats), L.ptr(Ks), width,
                                            height, eps2d, L.ptr(radii), L.ptr(conics), None, L.ptr(v_means2d), L.ptr(v_depths),
                                            L.ptr(v_conics), None, L.ptr(v_means), L.ptr(v_quats), L.ptr(v_scales),
                                            L.ptr(v_viewmats), L.stream()), "bds_project_bwd")
        g = ctx.needs_input_grad
        return (v_means if g[0] else None, v_quats if g[1] else None, v_scales if g[2] else None, v_viewmats, None, None,
                None, None, None, None, None, None)


def fully_fused_projection(means: Tensor, quats: Tensor, scales: Tensor, viewmats: Tensor, Ks: Tensor, width: int,
                           height: int, eps2d: float = 0.3, near_plane: float = 0.01, far_plane: float = 1e10,
                           radius_clip: float = 0.0, calc_compensations: bool = False):
    """-> radii [C,N] i32, means2d [C,N,2], depths [C,N], conics [C,N,3], compensations [C,N] | None."""
    N = means.shape[0]
    assert means.shape == (N, 3) and quats.shape == (N, 4) and scales.shape == (N, 3), (means.shape, quats.shape, scales.shape)
    Cn = viewmats.shape[0]
    assert viewmats.shape == (Cn, 4, 4) and Ks.shape == (Cn, 3, 3), (viewmats.shape, Ks.shape)
    out = _Projection.apply(means, quats, scales, viewmats, Ks, int(width), int(height), float(eps2d), float(near_plane),
                            float(far_plane), float(radius_clip), bool(calc_compensations))
    if calc_compensations:
        return out
    return (*out, None)


# --------------------------------------------------------------------------------------------
# tile intersection
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int, tile_height: int,
                want_isect_ids: bool = True, conics: Optional[Tensor] = None,
                opacities: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor], Tensor, Tensor]:
    """means2d [C,N,2], radii [C,N] i32, depths [C,N] ->
    tiles_per_gauss [C,N] i32, isect_ids [M] i64 | None, flatten_ids [M] i32, isect_offsets [C,th,tw] i32.

    (gsplat's isect_tiles + isect_offset_encode in one call: the offsets fall out of the ordering.)

    With ``conics`` [C,N,3] and ``opacities`` [C,N], (tile, Gaussian) pairs in which no pixel can reach
    alpha >= 1/255 are dropped (exact, conservative tile culling): the rendered image and all gradients
    are unchanged, only the lists get shorter.  Without them the lists are gsplat's bounding-square lists."""
    L.require_gpu(means2d, radii, depths)
    Cn, N = radii.shape
    dev = means2d.device
    means2d, depths = _f32c(means2d.detach()), _f32c(depths.detach())
    radii = radii.contiguous()
    assert (conics is None) == (opacities is None)
    if conics is not None:
        conics, opacities = _f32c(conics.detach()), _f32c(opacities.detach())
        assert conics.shape == (Cn, N, 3) and opacities.shape == (Cn, N)
    lib = L.lib()
    tiles_per_gauss = torch.empty(Cn, N, device=dev, dtype=torch.int32)
    ws_bytes = lib.bds_isect_prepare_workspace_bytes(Cn, N)
    ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
    m, nv = C.c_int64(0), C.c_int64(0)
    with L.timed("isect_prepare"):
        L.check(lib.bds_isect_prepare(Cn, N, L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(conics), L.ptr(opacities),
                                      tile_size, tile_width, tile_height,
                                      L.ptr(tiles_per_gauss), L.ptr(ws), ws_bytes, C.byref(m), C.byref(nv), 0, L.stream()),
                "bds_isect_prepare")
    M = int(m.value)
    flatten_ids = torch.empty(M, device=dev, dtype=torch.int32)
    isect_ids = torch.empty(M, device=dev, dtype=torch.int64) if want_isect_ids else None
    isect_offsets = torch.empty(Cn, tile_height, tile_width, device=dev, dtype=torch.int32)
    ws2_bytes = lib.bds_isect_build_workspace_bytes(Cn, N, M)
    ws2 = torch.empty(max(ws2_bytes, 16), device=dev, dtype=torch.uint8)
    with L.timed("isect_build"):
        L.check(lib.bds_isect_build(Cn, N, M, int(nv.value), L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(conics), L.ptr(opacities),
                                    tile_size, tile_width, tile_height,
                                    L.ptr(ws), ws_bytes, L.ptr(ws2), ws2_bytes, L.ptr(isect_ids), L.ptr(flatten_ids),
                                    L.ptr(isect_offsets), None, 0, L.stream()), "bds_isect_build")
    return tiles_per_gauss, isect_ids, flatten_ids, isect_offsets


# --------------------------------------------------------------------------------------------
# compositing
# --------------------------------------------------------------------------------------------
_BWD_SCHEDULE = os.environ.get("BDS_BWD_SCHEDULE", "1") != "0"


def set_bwd_schedule(on: bool) -> None:
    """Longest-tile-first launch order for the composite backward (include/bds.h: bds_rasterize_bwd_schedule).
    Affects timing only.  Default on; BDS_BWD_SCHEDULE=0 turns it off."""
    global _BWD_SCHEDULE
    _BWD_SCHEDULE = bool(on)


def bwd_schedule(C_: int, width: int, height: int, list_tile_size: int, isect_offsets: Tensor, last_ids: Tensor) -> Optional[Tensor]:
    """isect_offsets: the lists' offsets, built for tiles of ``list_tile_size`` px (a multiple of the 16-px compositing tile)."""
    if not _BWD_SCHEDULE:
        return None
    tw, th = math.ceil(width / TILE_SIZE), math.ceil(height / TILE_SIZE)
    order = torch.empty(int(L.lib().bds_rasterize_schedule_ints(C_, tw, th)), device=last_ids.device, dtype=torch.int32)
    L.check(L.lib().bds_rasterize_bwd_schedule(C_, width, height, TILE_SIZE, list_tile_size, tw, th, L.ptr(isect_offsets),
                                               L.ptr(last_ids), L.ptr(order), L.stream()), "bds_rasterize_bwd_schedule")
    return order


class _RasterizeToPixels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, isect_offsets, flatten_ids,
                absgrad):
        L.require_gpu(means2d, conics, colors, opacities, backgrounds)
        means2d_c, conics, colors, opacities, backgrounds = map(_f32c, (means2d, conics, colors, opacities, backgrounds))
        Cn, N = means2d_c.shape[0], means2d_c.shape[1]
        CH = colors.shape[-1]
        tw, th = math.ceil(width / TILE_SIZE), math.ceil(height / TILE_SIZE)   # compositing tiles; tile_size = the lists' tile
        assert isect_offsets.shape == (Cn, math.ceil(height / tile_size), math.ceil(width / tile_size)), (isect_offsets.shape, tile_size)
        M = flatten_ids.shape[0]
        dev = means2d_c.device
        lib, st = L.lib(), L.stream()
        # splat records in array order (record index = cam*N + g = what flatten_ids holds)
        rec = torch.empty(Cn * N, L.SPLAT_RECORD_FLOATS, device=dev, dtype=torch.float32)
        render = torch.empty(Cn, height, width, CH, device=dev, dtype=torch.float32)
        # (second plane: every pixel's final transmittance itself -- the backward starts from it, not from 1 - alpha; include/bds.h)
        a2 = torch.empty(2, Cn, height, width, 1, device=dev, dtype=torch.float32)
        alphas, t_final = a2[0], a2[1]
        last_ids = torch.empty(Cn, height, width, device=dev, dtype=torch.int32)
        with L.timed("rasterize_fwd"):
            L.check(lib.bds_splat_pack(Cn * N, CH, None, L.ptr(means2d_c), L.ptr(conics), L.ptr(colors), L.ptr(opacities), None, L.ptr(rec), st),
                    "bds_splat_pack")
            L.check(lib.bds_rasterize_fwd(Cn, Cn * N, M, CH, L.ptr(rec), L.ptr(backgrounds), width, height, TILE_SIZE, tile_size, tw, th,
                                          L.ptr(isect_offsets), L.ptr(flatten_ids), L.ptr(render), L.ptr(alphas), L.ptr(t_final), L.ptr(last_ids), st),
                    "bds_rasterize_fwd")
        ctx.save_for_backward(means2d, rec, backgrounds, isect_offsets, flatten_ids, alphas, last_ids, t_final)
        ctx.cfg = (width, height, tile_size, absgrad, CH)
        return render, alphas

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        means2d, rec, backgrounds, isect_offsets, flatten_ids, alphas, last_ids, t_final = ctx.saved_tensors
        width, height, tile_size, absgrad, CH = ctx.cfg
        Cn, N = means2d.shape[0], means2d.shape[1]
        tw, th = math.ceil(width / TILE_SIZE), math.ceil(height / TILE_SIZE)
        M = flatten_ids.shape[0]
        v_render, v_alphas = _f32c(v_render), _f32c(v_alphas)
        # gradient records (64 bytes per entry, accumulated with atomics): include/bds.h bds_rasterize_bwd
        v_rec = torch.zeros(Cn * N, L.GRAD_RECORD_FLOATS, device=rec.device, dtype=torch.float32)
        order = bwd_schedule(Cn, width, height, tile_size, isect_offsets, last_ids)
        with L.timed("rasterize_bwd"):
            L.check(L.lib().bds_rasterize_bwd(Cn, Cn * N, M, CH, L.ptr(rec), L.ptr(backgrounds), width, height, TILE_SIZE, tile_size, tw, th,
                                              L.ptr(isect_offsets), L.ptr(flatten_ids), L.ptr(alphas), L.ptr(t_final), L.ptr(last_ids), L.ptr(v_render),
                                              L.ptr(v_alphas), L.ptr(v_rec), int(bool(absgrad)), L.ptr(order), L.stream()),
                    "bds_rasterize_bwd")
        v = v_rec.view(Cn, N, L.GRAD_RECORD_FLOATS)
        v_colors, v_conics, v_means2d, v_opac = v[..., 0:CH], v[..., 4:7], v[..., 7:9], v[..., 11]
        if absgrad:
            # same contract as gsplat: the tensor the caller holds in meta["means2d"] grows `.absgrad`
            # (read at /root/reference/project/models/trainers/base.py:282)
            v_abs = v[..., 9:11].contiguous()
            prev = getattr(means2d, "absgrad", None)  # set by an earlier channel chunk of this backward
            means2d.absgrad = v_abs if prev is None else prev + v_abs
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[4]:
            v_bg = (v_render * (1.0 - alphas)).sum(dim=(1, 2))
        return v_means2d, v_conics, v_colors, v_opac, v_bg, None, None, None, None, None, None


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, image_width: int,
                        image_height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
                        backgrounds: Optional[Tensor] = None, absgrad: bool = False) -> Tuple[Tensor, Tensor]:
    """means2d [C,N,2], conics [C,N,3], colors [C,N,D], opacities [C,N] -> render [C,H,W,D], alphas [C,H,W,1].
    ``tile_size``: the tile size ``isect_offsets`` / ``flatten_ids`` were built for (gsplat's argument): 16, or a multiple of 16 --
    the compositor always works on 16 x 16 tiles and filters the candidates of a larger list tile per 16 x 16 tile (include/bds.h,
    "coarse lists").  As in gsplat, the result DOES depend on it slightly: through this operator the splat records carry no
    projection radius, so a Gaussian is clipped at the bounding square's LIST-tile granularity (tile_size 32 / 64 keep a little more
    of a splat's tail than 16; oracle-checked per tile size).  Only the fused view, whose records carry the radii, produces the
    16-px image for any list tile size."""
    Cn, N = means2d.shape[0], means2d.shape[1]
    assert means2d.shape == (Cn, N, 2) and conics.shape == (Cn, N, 3) and opacities.shape == (Cn, N), (
        means2d.shape, conics.shape, opacities.shape)
    assert colors.shape[:2] == (Cn, N), colors.shape
    assert tile_size >= TILE_SIZE and tile_size % TILE_SIZE == 0, f"tile_size {tile_size}: a multiple of {TILE_SIZE} is required"
    D = colors.shape[-1]
    if backgrounds is not None:
        assert backgrounds.shape == (Cn, D), backgrounds.shape
    if D in (1, 3, 4):
        return _RasterizeToPixels.apply(means2d, conics, colors, opacities, backgrounds, int(image_width), int(image_height),
                                        tile_size, isect_offsets, flatten_ids, absgrad)
    # other channel counts: split into the compiled widths (4, 3, 1); every chunk repeats the blend
    outs, alphas, s = [], None, 0
    while s < D:
        rem = D - s
        w = 4 if rem >= 4 else (3 if rem == 3 else 1)
        bg = None if backgrounds is None else backgrounds[..., s:s + w]
        r, alphas = _RasterizeToPixels.apply(means2d, conics, colors[..., s:s + w], opacities, bg, int(image_width),
                                             int(image_height), tile_size, isect_offsets, flatten_ids, absgrad)
        outs.append(r)
        s += w
    return torch.cat(outs, dim=-1), alphas
