"""``rasterization`` with the signature the reference imports from gsplat v1.3.0
(/root/reference/project/models/gaussians/basics.py:12; called at
/root/reference/project/models/trainers/base.py:393-408 and :811-826).

Pipeline (every stage a hand-written gfx950 kernel behind libbds.so):
    projection -> tile intersection + ordering -> alpha compositing -> expected-depth normalise
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

import weakref

from . import _lib as L
from .lazy_gaussians import lazy_source, materialised


def _tfinal_ptr(alphas):       # (the T_final plane behind an ``alphas`` tensor: fused_view._tfinal_ptr)
    from .fused_view import _tfinal_ptr as f
    return f(alphas)


_SPLIT_RENDER = os.environ.get("BDS_API_SPLIT_RENDER", "1") == "1"


class SplitRender(torch.Tensor):
    """``render_colors`` [..., H, W, 4] of the raw one-view node as a placeholder over its TWO outputs, rgb [..., H, W, 3] and depth
    [..., H, W, 1].  The reference's trainer takes ``renders[0]`` and splits it at once (``torch.split(renders, [3, 1], dim=-1)``,
    /root/reference/project/models/trainers/base.py:409-419); through a 4-channel tensor that costs a select backward, a slice backward
    (an image of zeros + a copy each) and a contiguous copy of the colours for the transform -- 0.13 ms of GPU and 0.15 ms of host time
    per 1080p view.  The placeholder answers exactly those uses with the node's own outputs (leading integer index, ``split([3, 1],
    -1)``, ``[..., :3]``, ``[..., 3:4]``, metadata); ANYTHING else first materialises ``torch.cat((rgb, depth), -1)`` -- an ordinary
    tensor with the same values and the same gradients (the mechanism of ``lazy_gaussians.LazyField``)."""

    @staticmethod
    def __new__(cls, rgb: Tensor, depth: Tensor, first=None):
        """``first``: (rgb[0], depth[0]) when the caller already holds them as tensors of their own (the node's outputs are [H,W,3] /
        [H,W,1]; the [1,H,W,.] form is an unsqueeze of them, whose backward is a view -- a select's is an image of zeros + a copy)."""
        assert rgb.shape[:-1] == depth.shape[:-1] and rgb.shape[-1] == 3 and depth.shape[-1] == 1
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(rgb.shape[:-1]) + (4,), dtype=rgb.dtype, device=rgb.device,
                                                requires_grad=rgb.requires_grad or depth.requires_grad)
        r._rgb, r._depth, r._cat, r._first = rgb, depth, None, first
        return r

    def materialise(self) -> Tensor:
        if self._cat is None:
            self._cat = torch.cat((self._rgb, self._depth), dim=-1)
        return self._cat

    def __repr__(self):
        return f"SplitRender(rgb={tuple(self._rgb.shape)}, depth={tuple(self._depth.shape)})"

    @staticmethod
    def _channel_slice(idx):
        """'rgb' / 'depth' for an index that selects channels 0:3 / 3:4 of the last dimension and everything else, else None."""
        if not (isinstance(idx, tuple) and len(idx) == 2 and idx[0] is Ellipsis and isinstance(idx[1], slice)):
            return None
        sl = idx[1]
        if sl.step not in (None, 1):
            return None
        lo, hi = sl.start or 0, 4 if sl.stop is None else sl.stop
        return "rgb" if (lo, hi) == (0, 3) else ("depth" if (lo, hi) == (3, 4) else None)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from .lazy_gaussians import _meta_funcs
        kwargs = kwargs or {}
        if func in _meta_funcs():
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        me = args[0] if args and isinstance(args[0], SplitRender) else None
        if me is not None and func is torch.Tensor.__getitem__ and len(args) == 2:
            idx = args[1]
            if isinstance(idx, int) and me.dim() > 3:                       # renders[0]
                if me._first is not None and idx in (0, -me.shape[0]) and me.shape[0] == 1:
                    return SplitRender(*me._first)
                return SplitRender(me._rgb[idx], me._depth[idx])
            which = cls._channel_slice(idx)
            if which is not None:
                return me._rgb if which == "rgb" else me._depth
            if isinstance(idx, tuple) and len(idx) >= 2 and isinstance(idx[0], int) and me.dim() > 3:   # renders[0, ..., :3]
                return me[idx[0]][idx[1:] if len(idx) > 2 else idx[1]]
        if me is not None and func in (torch.split, torch.Tensor.split):
            sizes = args[1] if len(args) > 1 else kwargs.get("split_size_or_sections", kwargs.get("split_size"))
            dim = args[2] if len(args) > 2 else kwargs.get("dim", 0)
            if list(sizes) == [3, 1] if isinstance(sizes, (list, tuple)) else False:
                if dim in (-1, me.dim() - 1):
                    return me._rgb, me._depth
        return func(*_split_materialised(args), **_split_materialised(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        return func(*_split_materialised(args), **_split_materialised(kwargs or {}))


def _split_materialised(x):
    if isinstance(x, SplitRender):
        return x.materialise()
    if isinstance(x, (list, tuple)):
        return type(x)(_split_materialised(v) for v in x)
    if isinstance(x, dict):
        return {k: _split_materialised(v) for k, v in x.items()}
    return x
from .gs_ops import (TILE_SIZE, _f32c, bwd_schedule, fully_fused_projection, isect_tiles, rasterize_to_pixels, spherical_harmonics)


class _Meta(dict):
    """The ``meta`` dict of gsplat's rasterization().  ``isect_ids`` (the sorted 64-bit keys, which nothing on
    the reference's path reads) is materialised on first access instead of on every step -- and, on the one-view fast path
    (``_RasterizeView``), so are gsplat's per-16-px-tile lists (``tiles_per_gauss`` / ``flatten_ids`` / ``isect_offsets``): the
    compositor there walks compact lists of 64-px tiles that never take gsplat's layout (the reference reads ``means2d`` / ``radii`` /
    ``width`` / ``height``: models/trainers/base.py:279-297,422-430)."""
    _LISTS = ("tiles_per_gauss", "flatten_ids", "isect_offsets")

    # (every reading accessor goes through __getitem__: meta.get(..), .items(), .values(), dict(meta) and ** unpacking see the lists too)
    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):      # (overridden on purpose: CPython's dict(meta) / ** take a raw-copy fast path for dict subclasses that keep dict's iterator)
        return iter(list(dict.keys(self)))

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __getitem__(self, key):
        if key in self._LISTS and dict.__getitem__(self, key) is None:
            cull = dict.__getitem__(self, "_cull")
            tpg, _, fids, offs = isect_tiles(self["means2d"], self["radii"], self["depths"], self["tile_size"], self["tile_width"],
                                             self["tile_height"], want_isect_ids=False, conics=self["conics"] if cull else None,
                                             opacities=self["opacities"].contiguous() if cull else None)
            dict.__setitem__(self, "tiles_per_gauss", tpg)
            dict.__setitem__(self, "flatten_ids", fids)
            dict.__setitem__(self, "isect_offsets", offs)
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


# One camera, post-activation colours [N,3], "RGB" / "RGB+ED", classic mode, no backgrounds -- the reference's two call patterns
# (models/trainers/base.py:393-408,811-826) -- run as ONE autograd node over compact lists (BDS_API_FUSED=0: the operator chain below)
_ONE_VIEW_NODE = os.environ.get("BDS_API_FUSED", "1") != "0"
_CHECK_FINITE = os.environ.get("BDS_API_CHECK_FINITE", "1") != "0"    # the raw one-view node's NaN / Inf check (vanilla.py:407-412)
_LIST_TILE = 64          # list tiles of the one-view node (the splat records carry the radii: the image is the 16-px one)
_CAPACITY: Dict[tuple, list] = {}      # (N, W, H) -> [list entries, visible Gaussians] seen so far: buffers provisioned before the wait


class _RasterizeView(torch.autograd.Function):
    """gsplat's rasterization() for one camera as one node: general projection (activated scales), visibility compaction + depth
    order + 64-px tile lists in compact positions (ONE host wait for the two list counts, overlapped with packing the colours),
    48-byte splat records of the visible Gaussians only, forward composite; backward: composite -> 64-byte gradient records of the
    visible Gaussians -> list-driven projection backward into dense, zero-initialised gradients (rows of culled Gaussians stay
    zero, as gsplat returns them).  Against the operator chain: no dense record pack / dense gradient records over all N, no dense
    projection backward, ~12 launches instead of ~45."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmat, Kmat, cfg):
        from .fused_view import _host_sync_objects, _release_sync_objects
        L.require_gpu(means, quats, scales, opacities, colors, viewmat, Kmat)
        means, quats, scales, opacities, colors, viewmat, Kmat = map(_f32c, (means, quats, scales, opacities, colors, viewmat, Kmat))
        lib, st = L.lib(), L.stream()
        dev = means.device
        W, H, N = cfg["width"], cfg["height"], means.shape[0]
        radii = torch.empty(1, N, device=dev, dtype=torch.int32)
        means2d, depths, conics = torch.empty(1, N, 2, device=dev), torch.empty(1, N, device=dev), torch.empty(1, N, 3, device=dev)
        with L.timed("project_fwd"):
            L.check(lib.bds_project_fwd(1, N, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(viewmat), L.ptr(Kmat), W, H, cfg["eps2d"],
                                        cfg["near_plane"], cfg["far_plane"], cfg["radius_clip"], L.ptr(radii), L.ptr(means2d), L.ptr(depths),
                                        L.ptr(conics), None, st), "bds_project_fwd")
        LT = _LIST_TILE
        ltw, lth = math.ceil(W / LT), math.ceil(H / LT)
        tw, th = math.ceil(W / TILE_SIZE), math.ceil(H / TILE_SIZE)
        opac_c = opacities.view(1, N)
        cull = cfg["cull"]
        cptr, optr = (L.ptr(conics), L.ptr(opac_c)) if cull else (None, None)
        tiles_per_gauss = torch.empty(1, N, device=dev, dtype=torch.int32)
        ws_bytes = lib.bds_isect_prepare_workspace_bytes(1, N)
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        counts, ev = _host_sync_objects(dev)
        with L.timed("isect_prepare"):
            L.check(lib.bds_isect_prepare_async(1, N, L.ptr(means2d), L.ptr(radii), L.ptr(depths), cptr, optr, LT, ltw, lth,
                                                L.ptr(tiles_per_gauss), L.ptr(ws), ws_bytes, counts.data_ptr(), ev.cuda_event, 1, st),
                    "bds_isect_prepare_async")
        # while the two counts travel to the host: every buffer whose size is known
        colors3 = colors.reshape(N, 3)
        key = (N, W, H)
        cap = _CAPACITY.setdefault(key, [0, 0])
        flat_buf = torch.empty(cap[0], device=dev, dtype=torch.int32) if cap[0] else None
        ws2_bytes = lib.bds_isect_build_workspace_bytes(1, N, cap[0]) if cap[0] else 0
        ws2 = torch.empty(max(ws2_bytes, 16), device=dev, dtype=torch.uint8) if cap[0] else None
        rec_buf = torch.empty(cap[1], L.SPLAT_RECORD_FLOATS, device=dev) if cap[1] else None
        isect_offsets = torch.empty(1, lth, ltw, device=dev, dtype=torch.int32)
        render = torch.empty(1, H, W, 4, device=dev)
        alphas = torch.empty(2, H, W, 1, device=dev)[0:1]     # (second plane: every pixel's final transmittance, fused_view._tfinal_ptr)
        last_ids = torch.empty(1, H, W, device=dev, dtype=torch.int32)
        ev.synchronize()
        M, n_vis = int(counts.np[0]), int(counts.np[1])
        _release_sync_objects(dev, (counts, ev))
        if flat_buf is None or M > cap[0]:
            flat_buf = torch.empty(M, device=dev, dtype=torch.int32)
            ws2_bytes = lib.bds_isect_build_workspace_bytes(1, N, M)
            ws2 = torch.empty(max(ws2_bytes, 16), device=dev, dtype=torch.uint8)
        if rec_buf is None or n_vis > cap[1]:
            rec_buf = torch.empty(n_vis, L.SPLAT_RECORD_FLOATS, device=dev)
        if M + M // 16 > cap[0]:
            cap[0] = M + M // 6 + 4096
        if n_vis + n_vis // 16 > cap[1]:
            cap[1] = n_vis + n_vis // 6 + 1024
        flatten, rec = flat_buf[:M], rec_buf[:n_vis]
        off = lib.bds_isect_visible_ids_offset(1, N)
        vis_ids = ws[off:off + 4 * n_vis].view(torch.int32)       # ascending ids of the visible Gaussians, read in place
        with L.timed("isect_build"):
            L.check(lib.bds_isect_build(1, N, M, n_vis, L.ptr(means2d), L.ptr(radii), L.ptr(depths), cptr, optr, LT, ltw, lth, L.ptr(ws),
                                        ws_bytes, L.ptr(ws2), ws2_bytes, None, L.ptr(flatten), L.ptr(isect_offsets), None, 1, st),
                    "bds_isect_build")
        with L.timed("rasterize_fwd"):
            L.check(lib.bds_splat_pack_rgbd(n_vis, L.ptr(vis_ids), L.ptr(means2d), L.ptr(conics), L.ptr(colors3), L.ptr(depths),
                                            L.ptr(opacities), L.ptr(radii), L.ptr(rec), st), "bds_splat_pack_rgbd")
            L.check(lib.bds_rasterize_fwd(1, n_vis, M, 4, L.ptr(rec), None, W, H, TILE_SIZE, LT, tw, th, L.ptr(isect_offsets), L.ptr(flatten),
                                          L.ptr(render), L.ptr(alphas), _tfinal_ptr(alphas), L.ptr(last_ids), st), "bds_rasterize_fwd")
        ctx.save_for_backward(means, quats, scales, opacities, viewmat, Kmat, rec, vis_ids, ws, flatten, isect_offsets, render, alphas, last_ids)
        ctx.cfg, ctx.M = cfg, M
        if cfg["ed"]:      # expected depth: D / clamp(alpha, 1e-10) (gsplat "ED")
            out = torch.empty_like(render)
            L.check(lib.bds_expected_depth_fwd(H * W, L.ptr(render), L.ptr(alphas), L.ptr(out), st), "bds_expected_depth_fwd")
        else:
            out = render[..., :3] if cfg["channels"] == 3 else render
        ctx.mark_non_differentiable(radii, depths, conics)
        # undefined output gradients stay None: the reference never puts a loss on meta["means2d"], and a materialised [1,N,2] zero
        # tensor would cost a fill, an index_select and an add over the records in every backward
        ctx.set_materialize_grads(False)
        return out, alphas, means2d, radii, depths, conics

    @staticmethod
    def backward(ctx, v_out, v_alphas, v_means2d_ext, *_):
        means, quats, scales, opacities, viewmat, Kmat, rec, vis_ids, _ws, flatten, isect_offsets, render, alphas, last_ids = ctx.saved_tensors
        cfg, M = ctx.cfg, ctx.M
        lib, st = L.lib(), L.stream()
        dev = means.device
        W, H, N = cfg["width"], cfg["height"], means.shape[0]
        n_vis = vis_ids.numel()
        tw, th = math.ceil(W / TILE_SIZE), math.ceil(H / TILE_SIZE)
        v_render, v_alphas_t = torch.empty_like(render), torch.empty_like(alphas)
        L.check(lib.bds_expected_depth_bwd(H * W, cfg["channels"], int(cfg["ed"]), L.ptr(render), L.ptr(alphas),
                                           None if v_out is None else L.ptr(_f32c(v_out)), None if v_alphas is None else L.ptr(_f32c(v_alphas)),
                                           L.ptr(v_render), L.ptr(v_alphas_t), st), "bds_expected_depth_bwd")
        v_alphas = v_alphas_t
        want_pose = bool(ctx.needs_input_grad[5])
        v_rec_all = torch.zeros(max(n_vis, 1) + (L.POSE_GRAD_SLOTS if want_pose else 0), L.GRAD_RECORD_FLOATS, device=dev)
        v_rec = v_rec_all[:max(n_vis, 1)]
        order = bwd_schedule(1, W, H, _LIST_TILE, isect_offsets, last_ids)
        with L.timed("rasterize_bwd"):
            L.check(lib.bds_rasterize_bwd(1, n_vis, M, 4, L.ptr(rec), None, W, H, TILE_SIZE, _LIST_TILE, tw, th, L.ptr(isect_offsets),
                                          L.ptr(flatten), L.ptr(alphas), _tfinal_ptr(alphas), L.ptr(last_ids), L.ptr(v_render), L.ptr(v_alphas), L.ptr(v_rec),
                                          int(bool(cfg["absgrad"])), L.ptr(order), st), "bds_rasterize_bwd")
        if v_means2d_ext is not None and n_vis:   # a loss term on meta["means2d"] itself: add its rows to the records
            v_rec[:n_vis, 7:9] += v_means2d_ext.reshape(N, 2).index_select(0, vis_ids.long())
        # ONE zero fill for all dense outputs: means 3 | quats 4 | scales 3 | opacities 1 | colours 3 | grad2d 2 | absgrad2d 2
        dense = torch.zeros(N * 18, device=dev)
        o = 0
        outs = []
        for w in (3, 4, 3, 1, 3, 2, 2):
            outs.append(dense[o:o + N * w].view(N, w) if w > 1 else dense[o:o + N])
            o += N * w
        v_means, v_quats, v_scales, v_opac, v_colors, g2d, ag2d = outs
        slots = v_rec_all[max(n_vis, 1):].view(L.POSE_GRAD_SLOTS, 4, 4) if want_pose else None
        with L.timed("project_bwd"):
            L.check(lib.bds_project_bwd_list(n_vis, L.ptr(vis_ids), L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(viewmat),
                                             L.ptr(Kmat), W, H, cfg["eps2d"], L.ptr(v_rec), L.ptr(v_means), L.ptr(v_quats), L.ptr(v_scales),
                                             L.ptr(v_opac), L.ptr(v_colors), L.ptr(slots), L.ptr(g2d), L.ptr(ag2d) if cfg["absgrad"] else None, st),
                    "bds_project_bwd_list")
        carrier = cfg["_means2d_ref"]() if cfg.get("_means2d_ref") is not None else None
        if carrier is not None:      # the tensor the caller holds in meta["means2d"] (trainers/base.py:280-297 read .absgrad / .grad)
            if cfg["absgrad"]:
                carrier.absgrad = ag2d.view(1, N, 2)
            if carrier.retains_grad:
                carrier.grad = g2d.view(1, N, 2)
        g = ctx.needs_input_grad
        return (v_means if g[0] else None, v_quats if g[1] else None, v_scales if g[2] else None, v_opac if g[3] else None,
                v_colors.view(cfg["colors_shape"]) if g[4] else None, slots.sum(0) if want_pose else None, None, None)


class _RasterizeRawView(torch.autograd.Function):
    """rasterization() for one camera over a class's RAW parameters (``lazy_gaussians``: the reference's unmodified call sequence
    with ``marshalling.install``): what ``_RasterizeView`` does, with sigmoid / exp / quaternion normalisation inside the projection
    kernel (bds_project_view_fwd), the SH colours (+0.5, clamp) evaluated by the record pack for the VISIBLE Gaussians only, straight
    from the class's two SH parameters (bds_splat_pack_sh_split: no concatenation, no dense SH pass over all N), and the backward
    list-driven into the raw parameters' gradients (bds_sh_view_bwd_list_split, bds_project_view_bwd_list).  The reference's NaN /
    Inf check (vanilla.py:407-412) is one streaming launch whose flag word arrives with the list counts."""

    @staticmethod
    def forward(ctx, means, quats, log_scales, logits, dc, rest, viewmat, Kmat, cfg):
        from .fused_view import _composite, _image_buffers, _view_front
        L.require_gpu(means, quats, log_scales, logits, dc, rest, viewmat, Kmat)
        means, quats, log_scales, logits, dc, rest, viewmat, Kmat = map(_f32c, (means, quats, log_scales, logits, dc, rest, viewmat, Kmat))
        lib, st = L.lib(), L.stream()
        dev = means.device
        W, H, N = cfg["width"], cfg["height"], means.shape[0]
        fcfg = dict(width=W, height=H, K=Kmat, cam_pos=cfg["cam_pos"], sh_degree=cfg["sh_degree"], near_plane=cfg["near_plane"],
                    far_plane=cfg["far_plane"], radius_clip=cfg["radius_clip"], eps2d=cfg["eps2d"], tile_cull=cfg["cull"],
                    list_tile=_LIST_TILE)
        flags = None
        if cfg["check_finite"]:
            # in FRONT of the projection: the flag's copy to the host is then older than the event the one wait of this view waits for.
            # On the raw values, by what makes each tensor's ACTIVATION non-finite (the reference checks the activated tensors,
            # vanilla.py:393-395,407-412): a log-scale whose exp overflows (NaN, +Inf, >= 88.72284; -Inf gives 0), a quaternion with a
            # NaN / Inf component or all zeros (0 / 0), a NaN logit (sigmoid maps +-Inf to 0 / 1); means and SH coefficients as they are
            flags = _finite_words(dev)
            ts = (means, quats, log_scales, logits, dc, rest)
            ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            cnts = (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])
            kinds = (ctypes.c_int * len(ts))(0, 2 if quats.data_ptr() % 16 == 0 else 0, 1, 3, 0, 0)
            L.check(lib.bds_nonfinite_flags_kinds(len(ts), ptrs, cnts, kinds, flags[0].data_ptr(), flags[1].data_ptr(), st), "bds_nonfinite_flags")
        f = _view_front(fcfg, means, quats, log_scales, logits.reshape(N), (dc, rest), viewmat, lambda: _image_buffers(W, H, dev))
        if flags is not None and int(flags[1][0]):
            bad = [n for i, n in enumerate(("means", "quats", "scales", "opacities", "features_dc", "features_rest")) if int(flags[1][0]) >> i & 1]
            raise ValueError(f"NaN / Inf detected in gaussian {', '.join(bad)} at step {cfg['step']}")
        rec, render, alphas, last_ids = _composite(f, f.opac, f.pre)
        ctx.save_for_backward(f.means, f.quats, f.scales, f.opac, viewmat, Kmat, rec, f.vis_ids, f.ws, f.flatten, f.isect_offsets, render, alphas,
                              last_ids, f.sh_rgb, f.cam_pos)
        ctx.cfg, ctx.M, ctx.shapes = cfg, f.M, (tuple(log_scales.shape), tuple(cfg["logits_shape"]), tuple(dc.shape), tuple(rest.shape))
        ctx.mark_non_differentiable(f.radii, f.depths, f.conics, f.opac)
        ctx.set_materialize_grads(False)
        if cfg.get("split"):       # rgb and depth as two outputs (SplitRender): 4-channel modes only
            rgb, depth = torch.empty(H, W, 3, device=dev), torch.empty(H, W, 1, device=dev)
            L.check(lib.bds_expected_depth_split_fwd(H * W, int(cfg["ed"]), L.ptr(render), L.ptr(alphas), L.ptr(rgb), L.ptr(depth), st),
                    "bds_expected_depth_split_fwd")
            return rgb, depth, alphas, f.means2d, f.radii, f.depths, f.conics, f.opac
        if cfg["ed"]:
            out = torch.empty_like(render)
            L.check(lib.bds_expected_depth_fwd(H * W, L.ptr(render), L.ptr(alphas), L.ptr(out), st), "bds_expected_depth_fwd")
        else:
            out = render[..., :3] if cfg["channels"] == 3 else render
        return out, None, alphas, f.means2d, f.radii, f.depths, f.conics, f.opac

    @staticmethod
    def backward(ctx, v_out, v_depth, v_alphas, v_means2d_ext, *_):
        (means, quats, scales, opac, viewmat, Kmat, rec, vis_ids, _ws, flatten, isect_offsets, render, alphas, last_ids, sh_rgb,
         cam_pos) = ctx.saved_tensors
        cfg, M = ctx.cfg, ctx.M
        lib, st = L.lib(), L.stream()
        dev = means.device
        W, H, N = cfg["width"], cfg["height"], means.shape[0]
        n_vis = vis_ids.numel()
        tw, th = math.ceil(W / TILE_SIZE), math.ceil(H / TILE_SIZE)
        v_render, v_alphas_t = torch.empty_like(render), torch.empty_like(alphas)
        if cfg.get("split"):
            L.check(lib.bds_expected_depth_split_bwd(H * W, int(cfg["ed"]), L.ptr(render), L.ptr(alphas),
                                                     None if v_out is None else L.ptr(_f32c(v_out)), None if v_depth is None else L.ptr(_f32c(v_depth)),
                                                     None if v_alphas is None else L.ptr(_f32c(v_alphas)), L.ptr(v_render), L.ptr(v_alphas_t), st),
                    "bds_expected_depth_split_bwd")
        else:
            L.check(lib.bds_expected_depth_bwd(H * W, cfg["channels"], int(cfg["ed"]), L.ptr(render), L.ptr(alphas),
                                               None if v_out is None else L.ptr(_f32c(v_out)), None if v_alphas is None else L.ptr(_f32c(v_alphas)),
                                               L.ptr(v_render), L.ptr(v_alphas_t), st), "bds_expected_depth_bwd")
        want_pose = bool(ctx.needs_input_grad[6])
        v_rec_all = torch.zeros(max(n_vis, 1) + (L.POSE_GRAD_SLOTS if want_pose else 0), L.GRAD_RECORD_FLOATS, device=dev)
        v_rec = v_rec_all[:max(n_vis, 1)]
        order = bwd_schedule(1, W, H, _LIST_TILE, isect_offsets, last_ids)
        with L.timed("rasterize_bwd"):
            L.check(lib.bds_rasterize_bwd(1, n_vis, M, 4, L.ptr(rec), None, W, H, TILE_SIZE, _LIST_TILE, tw, th, L.ptr(isect_offsets),
                                          L.ptr(flatten), L.ptr(alphas), _tfinal_ptr(alphas), L.ptr(last_ids), L.ptr(v_render), L.ptr(v_alphas_t), L.ptr(v_rec),
                                          int(bool(cfg["absgrad"])), L.ptr(order), st), "bds_rasterize_bwd")
        if v_means2d_ext is not None and n_vis:
            v_rec[:n_vis, 7:9] += v_means2d_ext.reshape(N, 2).index_select(0, vis_ids.long())
        ls_shape, lg_shape, dc_shape, rest_shape = ctx.shapes
        K = 1 + rest_shape[1]
        # ONE zero fill for all dense outputs: means 3 | quats 4 | log_scales 3 | logits 1 | dc 3 | rest 3 (K - 1) | grad2d 2 | absgrad2d 2
        widths = (3, 4, 3, 1, 3, 3 * (K - 1), 2, 2)
        dense = torch.zeros(N * sum(widths), device=dev)
        outs, o = [], 0
        for w in widths:
            outs.append(dense[o:o + N * w].view(N, w))
            o += N * w
        v_means, v_quats, v_ls, v_logits, v_dc, v_rest, g2d, ag2d = outs
        with L.timed("sh_bwd"):
            L.check(lib.bds_sh_view_bwd_list_split(n_vis, L.ptr(vis_ids), K, cfg["sh_degree"], L.ptr(means), L.ptr(cam_pos), L.ptr(sh_rgb), 1,
                                                   L.ptr(v_rec), L.ptr(v_dc), L.ptr(v_rest), 0, st), "bds_sh_view_bwd_list_split")
        slots = v_rec_all[max(n_vis, 1):].view(L.POSE_GRAD_SLOTS, 4, 4) if want_pose else None
        with L.timed("project_bwd"):
            L.check(lib.bds_project_view_bwd_list(n_vis, L.ptr(vis_ids), L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opac), L.ptr(viewmat),
                                                  L.ptr(Kmat), W, H, cfg["eps2d"], L.ptr(v_rec), L.ptr(v_means), L.ptr(v_quats), L.ptr(v_ls),
                                                  L.ptr(v_logits), L.ptr(slots), L.ptr(g2d), L.ptr(ag2d) if cfg["absgrad"] else None, None, 0,
                                                  st), "bds_project_view_bwd_list")
        carrier = cfg["_means2d_ref"]() if cfg.get("_means2d_ref") is not None else None
        if carrier is not None:
            if cfg["absgrad"]:
                carrier.absgrad = ag2d.view(1, N, 2)
            if carrier.retains_grad:
                carrier.grad = g2d.view(1, N, 2)
        g = ctx.needs_input_grad
        return (v_means if g[0] else None, v_quats if g[1] else None, v_ls.view(ls_shape) if g[2] else None,
                v_logits.view(lg_shape) if g[3] else None, v_dc.view(dc_shape) if g[4] else None, v_rest.view(rest_shape) if g[5] else None,
                slots.sum(0) if want_pose else None, None, None)


_FINITE_WORDS: Dict[torch.device, tuple] = {}


def _finite_words(dev):
    """(device word, page-locked word) of bds_nonfinite_flags, one pair per device (a view waits for its counts before the next one)."""
    w = _FINITE_WORDS.get(dev)
    if w is None:
        w = _FINITE_WORDS[dev] = (torch.zeros(1, device=dev, dtype=torch.int32), torch.zeros(1, dtype=torch.int32).pin_memory())
    return w


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
    src = lazy_source(means, quats, scales, opacities, colors)     # marshalling.install: a class's raw parameters behind placeholders
    raw_ok = (src is not None and _ONE_VIEW_NODE and viewmats.shape[0] == 1 and sh_degree is None and backgrounds is None
              and render_mode in ("RGB", "RGB+ED") and rasterize_mode == "classic" and means.shape[0] > 0 and src.features_rest.shape[1] >= 1)
    if not raw_ok:
        means, quats, scales, opacities, colors = materialised((means, quats, scales, opacities, colors))
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

    if raw_ok:
        cfg = dict(width=width, height=height, eps2d=float(eps2d), near_plane=float(near_plane), far_plane=float(far_plane),
                   radius_clip=float(radius_clip), ed=render_mode == "RGB+ED", channels=3 if render_mode == "RGB" else 4,
                   absgrad=bool(absgrad), cull=_TILE_CULLING, sh_degree=src.sh_degree, cam_pos=_f32c(src.cam_pos.detach().reshape(3)),
                   logits_shape=tuple(src.logits.shape), step=src.step, check_finite=_CHECK_FINITE,
                   split=_SPLIT_RENDER and render_mode != "RGB")
        out, depth1, alphas, means2d, radii, depths, conics, opac = _RasterizeRawView.apply(
            src.means, src.quats, src.log_scales, src.logits, src.features_dc, src.features_rest, viewmats[0], Ks[0], cfg)
        if depth1 is not None:      # the render as a placeholder over the node's two image outputs (SplitRender)
            out = SplitRender(out[None], depth1[None], first=(out, depth1))
        cfg["_means2d_ref"] = weakref.ref(means2d)
        tile_width, tile_height = math.ceil(width / float(tile_size)), math.ceil(height / float(tile_size))
        meta = _Meta({"camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
                      "opacities": opac.detach()[None, :], "tile_width": tile_width, "tile_height": tile_height,
                      "tiles_per_gauss": None, "isect_ids": None, "flatten_ids": None, "isect_offsets": None, "width": width,
                      "height": height, "tile_size": tile_size, "n_cameras": C, "_cull": _TILE_CULLING})
        return out, alphas, meta

    if (_ONE_VIEW_NODE and C == 1 and N > 0 and sh_degree is None and colors.shape[-1] == 3 and backgrounds is None
            and render_mode in ("RGB", "RGB+ED") and rasterize_mode == "classic"):
        cfg = dict(width=width, height=height, eps2d=float(eps2d), near_plane=float(near_plane), far_plane=float(far_plane),
                   radius_clip=float(radius_clip), ed=render_mode == "RGB+ED", channels=3 if render_mode == "RGB" else 4,
                   absgrad=bool(absgrad), cull=_TILE_CULLING, colors_shape=tuple(colors.shape))
        out, alphas, means2d, radii, depths, conics = _RasterizeView.apply(means, quats, scales, opacities, colors, viewmats[0], Ks[0], cfg)
        cfg["_means2d_ref"] = weakref.ref(means2d)      # the backward attaches .absgrad (and .grad, when retained) to THIS tensor object
        tile_width, tile_height = math.ceil(width / float(tile_size)), math.ceil(height / float(tile_size))
        meta = _Meta({"camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
                      "opacities": opacities.detach()[None, :], "tile_width": tile_width, "tile_height": tile_height,
                      "tiles_per_gauss": None, "isect_ids": None, "flatten_ids": None, "isect_offsets": None, "width": width,
                      "height": height, "tile_size": tile_size, "n_cameras": C, "_cull": _TILE_CULLING})
        return out, alphas, meta

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
        "_cull": _TILE_CULLING,
    })
    return render_colors, render_alphas, meta
