"""One autograd node for a whole view: activations -> projection -> SH (visible only) -> tile ordering ->
alpha compositing -> expected depth -> clamp + sky blend + multi-scale bilateral transform.

Same arithmetic, in the same order, as the reference's training forward
(/root/reference/project/models/gaussians/vanilla.py:378-414 -> models/trainers/base.py:385-432 ->
models/trainers/scene_graph.py:286-294,86-120) and as ``harness.render_view_staged`` (which strings the
individual autograd operators together and is tested equal to this node).  What changes is only the
host side: the ~60 framework nodes / ~100 launches of the staged formulation become one node that
issues ~25 launches of libbds.so kernels back to back, so the step is bound by the GPU, not by Python.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
import weakref
from typing import Dict, Optional, Sequence

import torch
from torch import Tensor

from . import _lib as L
from . import gs_ops as ops
from .bilagrid import _levels_struct

TILE = 16        # compositing tile (one wave64 per tile)
# Tile the depth-ordered lists are built for (include/bds.h "coarse lists"): 16 = gsplat's lists; 64 = one pair per (64-px tile,
# Gaussian), filtered per compositing tile as the chunks are staged -- several times fewer pairs to emit and sort.
LIST_TILE = int(os.environ.get("BDS_LIST_TILE", "64"))
# Opt-in, measured slower: SH colours evaluated inside the splat-record pack, for the visible Gaussians only and in list order
# (bds_splat_pack_sh), instead of a pass over all N Gaussians in front of the tile lists (bds_sh_view_fwd).  One thread per record
# gathering its own 192-byte coefficient row costs more than it saves: pack + forward composite 183 -> 237 us for 48 us of SH pass
# removed (profiles/r02x).  Needs 16-byte aligned coefficient rows (K * 3 % 4 == 0).
SH_IN_PACK = os.environ.get("BDS_SH_IN_PACK", "0") == "1"
# device-count form (nobody waits for the counts, so a dense SH pass hides behind nothing): the record pack evaluates the colours
SH_IN_PACK_DEV = os.environ.get("BDS_SH_IN_PACK_DEV", "1") == "1"


_PROJ_ROWS = os.environ.get("BDS_PROJ_ROWS", "1") == "1"   # (A/B: 0 = the projection's outputs as five separate arrays)


def _dp(t):
    """Device address of a tensor that may be a COLUMN of the projection's row block (fused_view._front_begin); None -> NULL."""
    return None if t is None else t.data_ptr()


def _empty(shape, dev, dtype=torch.float32):
    return torch.empty(shape, device=dev, dtype=dtype)


_SYNC: Dict[torch.device, list] = {}


def _host_sync_objects(dev):
    """(page-locked int64[3], event) used to read the two list counts back without draining the stream.  A front OWNS its pair from
    ``_front_begin`` until ``_front_finish`` has read the counts (``_release_sync_objects``): any number of fronts may be in flight
    (``render_classes`` next to a training view)."""
    free = _SYNC.setdefault(dev, [])
    if free:
        return free.pop()
    counts = torch.zeros(3, dtype=torch.int64).pin_memory()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))   # materialises the underlying hipEvent_t
    counts.np = counts.numpy()                  # (reading two host integers through tensor indexing costs ~10 us)
    return counts, ev


def _release_sync_objects(dev, pair) -> None:
    _SYNC.setdefault(dev, []).append(pair)


# The colour transform's backward leaves its last stage to the compositor's backward (bds_rasterize_bwd_ms) where views run one after
# the other: -1 launch, -38 us of transform for +10 us of compositor at 1080p (one-stream frame 847 -> 867 it/s).  A frame that runs
# the next view's forward next to this backward (graph_view.FrameGraph(overlap=True)) asks for the three-launch form instead
# (cfg["defer_epilogue"] = False): there the transform's memory-bound last stage hides behind the other stream's compositor for free,
# while folded into the compositor's backward it lengthens the frame's VALU-bound critical kernel (1006 -> 968 it/s).
_DEFER_EPILOGUE = os.environ.get("BDS_DEFER_EPILOGUE", "1") != "0"


class ListCapacity:
    """Capacities of one camera's intersection lists for the DEVICE-COUNT form of the view (include/bds.h "device-count forms"):
    no host read-back between the tile counting and the list build, every launch sized by these bounds, the actual counts read from
    device memory -- the form a hipGraph can hold (``graph_view.ViewGraph``).  ``counts`` (page-locked int64[3]: M, visible,
    overflow) is written by the GPU during the view; the host looks at it after the fact (``observed`` / ``overflowed``)."""

    def __init__(self, m_cap: int, nvis_cap: int):
        self.m_cap, self.nvis_cap = int(m_cap), int(nvis_cap)
        self.counts = torch.zeros(3, dtype=torch.int64).pin_memory()
        self.counts.np = self.counts.numpy()

    def observed(self):
        """(M, visible) of the last completed visit (the caller has synchronised with it).  When the VISIBLE count outgrew its
        capacity the tile counting did not run (its launches are sized by that capacity) and M reads 0; ``wanted`` extrapolates."""
        return int(self.counts.np[0]), int(self.counts.np[1])

    def wanted(self):
        """(M, visible) to provision for after the last visit: ``observed()``; after a visible-count overflow (M unknown) the
        current list capacity scaled by the growth of the visible set."""
        M, n_vis = self.observed()
        if n_vis > self.nvis_cap:
            M = max(M, int(self.m_cap * (n_vis / max(self.nvis_cap, 1))) + 1)
        return M, n_vis

    def overflowed(self) -> bool:
        return bool(self.counts.np[2])


_LIST_CAPACITY: Dict[tuple, int] = {}   # (N, W, H, culling) -> entries to provision for the intersection lists


class _Info(dict):
    """info dict of the fused view.  The per-tile lists are kept as positions in ``visible_ids`` (the ascending ids of the visible
    Gaussians; ``flatten_ranks`` addresses the compact splat / gradient records); gsplat's ``flatten_ids`` (Gaussian ids) is
    derived on first access: ``visible_ids[flatten_ranks]``."""

    def __getitem__(self, k):
        if k == "flatten_ids" and not dict.__contains__(self, k):
            dict.__setitem__(self, k, dict.__getitem__(self, "visible_ids")[dict.__getitem__(self, "flatten_ranks").long()])
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k == "flatten_ids" or dict.__contains__(self, k)


class _Out(dict):
    """Result dict of the fused view.  Two of the reference trainer's outputs are only used for inspection / the (zero-weighted)
    cycle term and are materialised on first access, detached: ``rgb_gaussians`` = ``clamp(rendered_rgb, max=1.0)``
    (trainers/base.py:414; the transform consumes the clamp fused) and ``original_rgb`` = the colour entering the transform,
    ``rgb_gaussians + rgb_sky * (1 - opacity)`` (trainers/base.py:496-498)."""
    _LAZY = ("rgb_gaussians", "original_rgb")

    def __getitem__(self, k):
        if k in self._LAZY and not dict.__contains__(self, k):
            g = dict.__getitem__(self, "_rgb_g_raw").detach().clamp(max=1.0)
            if k == "original_rgb":
                g = g + dict.__getitem__(self, "_sky").detach() * (1.0 - dict.__getitem__(self, "opacity").detach())
            dict.__setitem__(self, k, g)
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k in self._LAZY or dict.__contains__(self, k)


class _Front:
    """One view up to its per-tile lists -- everything that does not depend on the opacities' VALUES beyond the tile cull: activations,
    projection, SH colours, depth order, per-tile lists (compact positions) and the ascending visible-id list.  Shared by the training
    forward and by the evaluation re-renders (``render_classes``), which composite several opacity masks over ONE front.
    ``_front_begin`` fills the first half (nothing there depends on the host), ``_front_finish`` waits for the two list counts and
    adds the lists.  Plain attribute bag: inputs (means, quats, log_scales, sh, viewmat, cam_pos), projection outputs (scales, opac,
    radii, means2d, depths, conics), colours (sh_rgb, colors / sh_by_rank), tile-stage state (ws, counts, ev, capacities ...) and,
    after ``_front_finish``: flatten, vis_ids, isect_offsets, M, n_vis."""
    pass


_VIS_CAPACITY: Dict[tuple, int] = {}   # visible-Gaussian count seen per configuration: the splat records are provisioned before the wait


SPLIT_POOL_PER_TILE = int(os.environ.get("BDS_SPLIT_POOL_PER_TILE", "4096"))   # refined-list pool: int32 words per long tile of the capacity (0 = no refinement)


def _split(cfg: dict, list_tile: int, n_tiles: int, have_schedule: bool):
    """(split_len, split_cap, split_pool) of the device-count compositors (include/bds.h): tiles whose list holds at least split_len
    entries -- at most split_cap of them -- are composited by four waves, strip by strip, over their own REFINED candidate lists (a
    pool of split_pool words behind the schedule).  (0, 0, 0) = off.  ``cfg["split_len"]`` / ``cfg["split_cap"]`` (graph_view sets them
    per view slot from the calibration visit's lists) or ``BDS_SPLIT_LEN`` / ``BDS_SPLIT_CAP`` (A/B sessions, tests; default capacity:
    every tile).  Needs the backward's schedule buffer (the long-tile list lives behind it): a forward without one is not split."""
    if list_tile <= TILE or not have_schedule:
        return 0, 0, 0
    n = int(cfg.get("split_len") or os.environ.get("BDS_SPLIT_LEN", "0"))
    if n <= 0:
        return 0, 0, 0
    cap = int(cfg.get("split_cap") or os.environ.get("BDS_SPLIT_CAP", "0")) or n_tiles
    cap = max(1, min(cap, n_tiles))
    return n, cap, min(cap * SPLIT_POOL_PER_TILE, (1 << 31) - 1)


_TILE_OPTIONS_READ = False


def _tile_stage_options() -> None:
    """A/B sessions: ``BDS_CAP_LAUNCH=0`` sizes the device-count tile stage's launches by N again (include/bds.h option 0)."""
    global _TILE_OPTIONS_READ
    if not _TILE_OPTIONS_READ:
        _TILE_OPTIONS_READ = True
        if os.environ.get("BDS_CAP_LAUNCH") is not None:
            L.set_option(L.OPT_CAP_LAUNCH, int(os.environ["BDS_CAP_LAUNCH"]))


def _view_front(cfg: dict, means, quats, log_scales, logits, sh, viewmat, before_wait=None) -> _Front:
    """``before_wait`` (optional callable): host work of the caller that does not depend on the list counts (allocations, level
    structs); it runs while the GPU is still producing them, so that after the one host wait of a view only launches remain."""
    return _front_finish(_front_begin(cfg, means, quats, log_scales, logits, sh, viewmat), before_wait)


def _front_begin(cfg: dict, means, quats, log_scales, logits, sh, viewmat) -> _Front:
    """First half of a view's forward: activations + projection, visibility compaction, depth order, tile counts (whose totals travel
    to the host asynchronously) and the SH colours.  Nothing here depends on the host."""
    sh_rest = None
    if isinstance(sh, (tuple, list)):     # split storage, as the reference's classes hold it: (band 0 [N,3], bands 1.. [N,K-1,3])
        sh, sh_rest = sh[0].contiguous(), sh[1].contiguous()
        assert sh.dim() == 2 and sh.shape[1] == 3 and sh_rest.dim() == 3 and sh_rest.shape[2] == 3 and cfg.get("caps") is None
    L.require_gpu(means, quats, log_scales, logits, sh, viewmat)
    lib, st = L.lib(), L.stream()
    dev = means.device
    W, H = cfg["width"], cfg["height"]
    N, K = means.shape[0], (sh.shape[1] if sh_rest is None else 1 + sh_rest.shape[1])
    means, quats, log_scales, logits, sh = (t.contiguous() for t in (means, quats, log_scales, logits, sh))
    viewmat, Kmat = viewmat.contiguous(), cfg["K"].contiguous()
    # activations (vanilla.py:393-394) + projection (C = 1)
    scales, opac = _empty((N, 3), dev), _empty((N,), dev)
    radii = _empty((1, N), dev, torch.int32)
    caps = cfg.get("caps")                 # ListCapacity: the device-count form (no host wait in this view)
    in_pack = cfg.get("sh_in_pack", SH_IN_PACK if caps is None else SH_IN_PACK_DEV)
    pack_colours = sh_rest is not None or (in_pack and (K * 3) % 4 == 0 and sh.data_ptr() % 16 == 0)
    if pack_colours and _PROJ_ROWS and N > 0:
        # the projection's outputs as the columns of ONE [N,8] block of 32-byte rows (csrc/bds_common.h ProjLayout): the tile stage
        # and the record pack gather one line per visible Gaussian instead of one per array; radii / opacities stay dense as well
        rows = _empty((N, 8), dev)
        means2d, depths, conics, opac_row = rows[None, :, 0:2], rows[None, :, 2], rows[None, :, 4:7], rows[:, 7]
    else:
        means2d, depths, conics, opac_row = _empty((1, N, 2), dev), _empty((1, N), dev), _empty((1, N, 3), dev), opac
    tiles_per_gauss = _empty((1, N), dev, torch.int32)
    ws_bytes = lib.bds_isect_prepare_workspace_bytes(1, N)
    ws = cfg.get("prep_ws")   # a caller-owned prepare workspace (graph_view: the lists and their counts outlive the view)
    if ws is None:
        ws = _empty((max(ws_bytes, 16),), dev, torch.uint8)
    assert ws.dtype == torch.uint8 and ws.is_contiguous() and ws.numel() >= ws_bytes and ws.device == dev
    # device-count form: the projection also does the tile stage's first launch (visible counts per workgroup, tables cleared)
    pre_reduced = False
    bounds = cfg.get("block_bounds")      # [cdiv(N, 256), 8] of bds_gaussian_block_bounds over THESE means / log_scales (graph_view: once per frame)
    with L.timed("project_fwd"):
        if cfg.get("caps") is not None and _PROJECT_PREPARES and N > 0:
            args = (N, L.ptr(means), L.ptr(quats), L.ptr(log_scales), L.ptr(logits), L.ptr(viewmat),
                    L.ptr(Kmat), W, H, cfg["eps2d"], cfg["near_plane"], cfg["far_plane"], cfg["radius_clip"],
                    L.ptr(scales), L.ptr(opac), L.ptr(radii), _dp(means2d), _dp(depths), _dp(conics),
                    L.ptr(tiles_per_gauss), L.ptr(ws), ws_bytes)
            rc = (lib.bds_project_view_prepare_fwd(*args, st) if bounds is None
                  else lib.bds_project_view_prepare_fwd_blocks(*args, L.ptr(bounds), st))
            pre_reduced = rc == L.BDS_OK
            if rc not in (L.BDS_OK, L.BDS_ECAPACITY):     # (ECAPACITY: N beyond the short sort path -- the plain projection below)
                L.check(rc, "bds_project_view_prepare_fwd")
        if not pre_reduced:
            args = (N, L.ptr(means), L.ptr(quats), L.ptr(log_scales), L.ptr(logits), L.ptr(viewmat),
                    L.ptr(Kmat), W, H, cfg["eps2d"], cfg["near_plane"], cfg["far_plane"], cfg["radius_clip"],
                    L.ptr(scales), L.ptr(opac), L.ptr(radii), _dp(means2d), _dp(depths), _dp(conics))
            L.check(lib.bds_project_view_fwd(*args, st) if bounds is None or N == 0 else lib.bds_project_view_fwd_blocks(*args, L.ptr(bounds), st),
                    "bds_project_view_fwd")
    # tile ordering
    LT = cfg.get("list_tile", LIST_TILE)
    tw, th = math.ceil(W / LT), math.ceil(H / LT)      # list tiles
    cull = cfg["tile_cull"]
    isect_offsets = _empty((1, th, tw), dev, torch.int32)
    cptr, optr = (_dp(conics), _dp(opac_row)) if cull else (None, None)
    if caps is not None:
        counts, ev = caps.counts, None
        _tile_stage_options()
        with L.timed("isect_prepare"):
            L.check(lib.bds_isect_prepare_dev(1, N, _dp(means2d), L.ptr(radii), _dp(depths), cptr, optr, LT, tw, th,
                                              L.ptr(tiles_per_gauss), L.ptr(ws), ws_bytes, caps.m_cap, caps.nvis_cap,
                                              counts.data_ptr(), 3 if pre_reduced else 1, st), "bds_isect_prepare_dev")
    else:
        counts, ev = _host_sync_objects(dev)
        with L.timed("isect_prepare"):
            L.check(lib.bds_isect_prepare_async(1, N, _dp(means2d), L.ptr(radii), _dp(depths), cptr, optr, LT, tw, th,
                                                L.ptr(tiles_per_gauss), L.ptr(ws), ws_bytes, counts.data_ptr(), ev.cuda_event, 1, st),
                    "bds_isect_prepare_async")
    # While the host waits for the two counts, the GPU evaluates the SH colours (vanilla.py:384-389), which do not
    # depend on the lists; the list buffers are provisioned beforehand from the largest count seen so far.
    cam_pos = cfg["cam_pos"].contiguous()
    if pack_colours:
        sh_rgb, colors = None, None        # evaluated by the record pack, for the visible Gaussians only (_composite)
    else:
        sh_rgb, colors = _empty((N, 3), dev), _empty((1, N, 4), dev)
        if caps is None or os.environ.get("BDS_SH_BEFORE_BUILD") == "1":
            with L.timed("sh_fwd"):
                L.check(lib.bds_sh_view_fwd(N, K, cfg["sh_degree"], L.ptr(means), L.ptr(cam_pos), L.ptr(sh), L.ptr(radii), L.ptr(depths),
                                            L.ptr(sh_rgb), L.ptr(colors), st), "bds_sh_view_fwd")
        # (device-count form: nobody waits, and the SH pass is enqueued BEHIND the list build -- _front_finish_dev: next to another
        # stream's compositor backward the build's five small launches crawl, a streaming pass over the coefficients does not)
    key = (N, W, H, bool(cull), LT)
    cap = _LIST_CAPACITY.get(key, 0) if caps is None else caps.m_cap
    buf, ws2, ws2_bytes = None, None, 0
    if cap:
        buf = _empty((cap,), dev, torch.int32)
        ws2_bytes = lib.bds_isect_build_workspace_bytes(1, N, cap)
        ws2 = _empty((max(ws2_bytes, 16),), dev, torch.uint8)
    vcap = _VIS_CAPACITY.get(key, 0) if caps is None else caps.nvis_cap
    rec_buf = _empty((vcap, L.SPLAT_RECORD_FLOATS), dev) if vcap else None
    off = lib.bds_isect_visible_ids_offset(1, N)
    f = _Front()
    f.cfg = cfg
    f.means, f.quats, f.log_scales, f.sh, f.viewmat, f.cam_pos = means, quats, log_scales, sh, viewmat, cam_pos
    f.sh_rest, f.K = sh_rest, K
    f.scales, f.opac, f.radii, f.means2d, f.depths, f.conics = scales, opac, radii, means2d, depths, conics
    f.opac_row = opac_row      # (the activated opacities where the tile stage / the pack read them: the rows' column, or `opac` itself)
    f.sh_rgb, f.colors, f.sh_by_rank, f.sh_degree = sh_rgb, colors, False, cfg["sh_degree"]
    f.tiles_per_gauss, f.isect_offsets, f.ws, f.ws_bytes, f.cull = tiles_per_gauss, isect_offsets, ws, ws_bytes, cull
    f.counts, f.ev, f.key, f.cap, f.vcap, f.caps = counts, ev, key, cap, vcap, caps
    f.buf, f.ws2, f.ws2_bytes, f.rec_buf, f.ids_offset = buf, ws2, ws2_bytes, rec_buf, off
    f.list_tile, f.list_tw, f.list_th = LT, tw, th
    f.W, f.H, f.N = W, H, N
    f.tw, f.th = math.ceil(W / TILE), math.ceil(H / TILE)   # compositing tiles
    return f


def _front_finish(f: _Front, before_wait=None) -> _Front:
    """Second half: the one host wait of a view (list counts), then the per-tile lists."""
    lib, st = L.lib(), L.stream()
    dev, N = f.means.device, f.N
    cptr, optr = (_dp(f.conics), _dp(f.opac_row)) if f.cull else (None, None)
    f.pre = before_wait() if before_wait is not None else None
    if f.caps is not None:
        return _front_finish_dev(f)
    f.ev.synchronize()
    M, n_vis = int(f.counts.np[0]), int(f.counts.np[1])
    _release_sync_objects(dev, (f.counts, f.ev))
    if f.rec_buf is None or n_vis > f.vcap:
        f.rec_buf = _empty((n_vis, L.SPLAT_RECORD_FLOATS), dev)
    if n_vis + n_vis // 16 > f.vcap:
        _VIS_CAPACITY[f.key] = n_vis + n_vis // 6 + 1024
    buf, ws2, ws2_bytes = f.buf, f.ws2, f.ws2_bytes
    if M > f.cap or buf is None:   # first call of this configuration, or the lists outgrew the expectation
        buf = _empty((M,), dev, torch.int32)
        ws2_bytes = lib.bds_isect_build_workspace_bytes(1, N, M)
        ws2 = _empty((max(ws2_bytes, 16),), dev, torch.uint8)
    f.flatten = buf[:M]                                # per-tile lists of COMPACT positions (they address the splat records)
    # ascending ids of the visible Gaussians: compact position -> id, the work list of everything downstream (walks memory in
    # order).  Read in place from the prepare workspace (which this view keeps alive): no copy node between the kernels.
    f.vis_ids = f.ws[f.ids_offset:f.ids_offset + 4 * n_vis].view(torch.int32)
    with L.timed("isect_build"):
        L.check(lib.bds_isect_build(1, N, M, n_vis, _dp(f.means2d), L.ptr(f.radii), _dp(f.depths), cptr, optr, f.list_tile, f.list_tw,
                                    f.list_th, L.ptr(f.ws), f.ws_bytes, L.ptr(ws2), ws2_bytes, None, L.ptr(f.flatten),
                                    L.ptr(f.isect_offsets), None, 1, st), "bds_isect_build")
    if M + M // 16 > f.cap:
        _LIST_CAPACITY[f.key] = M + M // 6 + 4096
    f.buf = f.ws2 = None
    f.M, f.n_vis = M, n_vis
    f.m_dev = f.nvis_dev = None
    return f


def _front_finish_dev(f: _Front) -> _Front:
    """Device-count form of the second half: no wait.  The lists are built into buffers of the capacities, every kernel downstream
    reads the two EFFECTIVE counts from the first words of the prepare workspace (f.m_dev / f.nvis_dev: device addresses)."""
    lib, st = L.lib(), L.stream()
    N = f.N
    cptr, optr = (_dp(f.conics), _dp(f.opac_row)) if f.cull else (None, None)
    M, n_vis = f.caps.m_cap, f.caps.nvis_cap
    f.flatten = f.buf
    f.vis_ids = f.ws[f.ids_offset:f.ids_offset + 4 * n_vis].view(torch.int32)
    with L.timed("isect_build"):
        L.check(lib.bds_isect_build_dev(1, N, M, n_vis, _dp(f.means2d), L.ptr(f.radii), _dp(f.depths), cptr, optr, f.list_tile,
                                        f.list_tw, f.list_th, L.ptr(f.ws), f.ws_bytes, L.ptr(f.ws2), f.ws2_bytes, L.ptr(f.flatten),
                                        L.ptr(f.isect_offsets), 1, st), "bds_isect_build_dev")
    if f.colors is not None and os.environ.get("BDS_SH_BEFORE_BUILD") != "1":
        with L.timed("sh_fwd"):
            L.check(lib.bds_sh_view_fwd(N, f.sh.shape[1], f.sh_degree, L.ptr(f.means), L.ptr(f.cam_pos), L.ptr(f.sh), L.ptr(f.radii),
                                        L.ptr(f.depths), L.ptr(f.sh_rgb), L.ptr(f.colors), st), "bds_sh_view_fwd")
    f.buf = f.ws2 = None
    f.M, f.n_vis = M, n_vis
    base = f.ws.data_ptr()
    f.m_dev, f.nvis_dev = base + lib.bds_isect_counts_offset(2), base + lib.bds_isect_counts_offset(3)
    return f


def _image_buffers(W: int, H: int, dev):
    """(render, alphas, last_ids) of one camera.  ``alphas`` is the first plane of a [2, H, W, 1] buffer: the second receives every
    pixel's final transmittance itself (``_tfinal_ptr``; include/bds.h bds_rasterize_fwd ``t_final``) -- the backward starts from that
    value instead of 1 - alpha, which has lost the low bits of a small transmittance."""
    a2 = _empty((2, H, W, 1), dev)
    return _empty((1, H, W, 4), dev), a2[0:1], _empty((1, H, W), dev, torch.int32)


def _tfinal_ptr(alphas: Tensor) -> int:
    """Device address of the T_final plane behind an ``alphas`` tensor made by ``_image_buffers``."""
    assert alphas.is_contiguous() and alphas.untyped_storage().nbytes() >= alphas.storage_offset() * 4 + 2 * alphas.numel() * 4
    return alphas.data_ptr() + alphas.numel() * 4


def _composite(f: _Front, opac: Tensor, images=None, zero_grad_records: Optional[Tensor] = None, tile_order: Optional[Tensor] = None,
               zero_tail: Optional[Tensor] = None):
    """Splat records of the visible Gaussians with the given opacities [N] + the forward composite (RGB + depth).
    ``images``: (render, alphas, last_ids) allocated by the caller (before the host wait), else allocated here.
    ``zero_grad_records`` (device-count form): [n_vis capacity + pose slots, 16] gradient records the pack clears on its way."""
    lib, st = L.lib(), L.stream()
    dev = opac.device
    n_vis, M, W, H = f.n_vis, f.M, f.W, f.H
    if opac is f.opac:          # (the view's own opacities: read from the rows; a masked copy -- render_classes -- is a dense array)
        opac = f.opac_row
    if f.m_dev is not None:
        if f.rec_buf is not None:
            rec, f.rec_buf = f.rec_buf[:n_vis], None
        else:
            rec = _empty((n_vis, L.SPLAT_RECORD_FLOATS), dev)
        render, alphas, last_ids = images if images is not None else _image_buffers(W, H, dev)
        zr = zero_grad_records
        tail = zero_tail if zero_tail is not None else (None if zr is None or zr.shape[0] <= n_vis else zr[n_vis:])
        with L.timed("rasterize_fwd"):
            if f.colors is None:   # SH colours evaluated by the pack (visible Gaussians only); un-clamped values kept in list order
                f.sh_rgb, f.sh_by_rank = _empty((max(n_vis, 1), 3), dev), True
                if f.cfg.get("row_catchup") is not None:
                    # a row-lazy optimizer (optim.DeferredRowAdam) brings the coefficient rows of THIS view's visible Gaussians up to
                    # its current step before the pack reads them: (capacity, device count, device id list)
                    f.cfg["row_catchup"](n_vis, f.nvis_dev, L.ptr(f.vis_ids))
                L.check(lib.bds_splat_pack_sh_dev(n_vis, f.nvis_dev, L.ptr(f.vis_ids), f.sh.shape[1], f.sh_degree, L.ptr(f.means),
                                                  L.ptr(f.cam_pos), L.ptr(f.sh), _dp(f.means2d), _dp(f.conics), _dp(f.depths),
                                                  _dp(opac), L.ptr(f.radii), L.ptr(rec), L.ptr(f.sh_rgb), L.ptr(zr), L.ptr(tail),
                                                  0 if tail is None else tail.numel(), L.ptr(tile_order), st), "bds_splat_pack_sh_dev")
            else:
                L.check(lib.bds_splat_pack_dev(n_vis, f.nvis_dev, 4, L.ptr(f.vis_ids), L.ptr(f.means2d), L.ptr(f.conics), L.ptr(f.colors),
                                               L.ptr(opac), L.ptr(f.radii), L.ptr(rec), L.ptr(zr), L.ptr(tail),
                                               0 if tail is None else tail.numel(), L.ptr(tile_order), st), "bds_splat_pack_dev")
            # (tile_order: the backward's schedule is left by the compositing waves themselves; its header was cleared by the pack)
            L.check(lib.bds_rasterize_fwd_dev(1, n_vis, M, f.m_dev, 4, L.ptr(rec), None, W, H, TILE, f.list_tile, f.tw, f.th,
                                              L.ptr(f.isect_offsets), L.ptr(f.flatten), L.ptr(render), L.ptr(alphas), _tfinal_ptr(alphas),
                                              L.ptr(last_ids), L.ptr(tile_order), *_split(f.cfg, f.list_tile, f.tw * f.th, tile_order is not None), st),
                    "bds_rasterize_fwd_dev")
        return rec, render, alphas, last_ids
    if f.rec_buf is not None:      # provisioned before the wait (first composite over this front only)
        rec, f.rec_buf = f.rec_buf[:n_vis], None
    else:
        rec = _empty((n_vis, L.SPLAT_RECORD_FLOATS), dev)
    render, alphas, last_ids = images if images is not None else _image_buffers(W, H, dev)
    with L.timed("rasterize_fwd"):
        if f.colors is None:       # SH colours evaluated on the way; their un-clamped values stay in list order for the backward
            f.sh_rgb, f.sh_by_rank = _empty((max(n_vis, 1), 3), dev), True
            if f.sh_rest is not None:
                L.check(lib.bds_splat_pack_sh_split(n_vis, L.ptr(f.vis_ids), f.K, f.sh_degree, L.ptr(f.means), L.ptr(f.cam_pos), L.ptr(f.sh),
                                                    L.ptr(f.sh_rest), _dp(f.means2d), _dp(f.conics), _dp(f.depths), _dp(opac),
                                                    L.ptr(f.radii), L.ptr(rec), L.ptr(f.sh_rgb), st), "bds_splat_pack_sh_split")
            else:
                L.check(lib.bds_splat_pack_sh(n_vis, L.ptr(f.vis_ids), f.sh.shape[1], f.sh_degree, L.ptr(f.means), L.ptr(f.cam_pos),
                                              L.ptr(f.sh), _dp(f.means2d), _dp(f.conics), _dp(f.depths), _dp(opac), L.ptr(f.radii),
                                              L.ptr(rec), L.ptr(f.sh_rgb), st), "bds_splat_pack_sh")
        else:
            L.check(lib.bds_splat_pack(n_vis, 4, L.ptr(f.vis_ids), L.ptr(f.means2d), L.ptr(f.conics), L.ptr(f.colors), L.ptr(opac),
                                       L.ptr(f.radii), L.ptr(rec), st), "bds_splat_pack")
        L.check(lib.bds_rasterize_fwd(1, n_vis, M, 4, L.ptr(rec), None, W, H, TILE, f.list_tile, f.tw, f.th, L.ptr(f.isect_offsets), L.ptr(f.flatten),
                                      L.ptr(render), L.ptr(alphas), _tfinal_ptr(alphas), L.ptr(last_ids), st), "bds_rasterize_fwd")
    return rec, render, alphas, last_ids


class _FusedView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg: dict, means, quats, log_scales, logits, sh, sky, viewmat, *grids):
        steps = _FusedView.forward_steps(ctx, cfg, means, quats, log_scales, logits, sh, sky, viewmat, *grids)
        next(steps)                      # Gaussian half: projection, lists, SH, compositor
        try:
            next(steps)                  # image half: expected depth, clamp, sky blend, bilateral transform
        except StopIteration as done:
            return done.value
        raise AssertionError("forward_steps yields once")

    @staticmethod
    def forward_steps(ctx, cfg: dict, means, quats, log_scales, logits, sh, sky, viewmat, *grids):
        """The forward as a generator that yields ONCE (the projection's radii), between the compositor and the colour transform:
        ``graph_view`` may replay the second half -- with the loss and the backward's image half -- on the backward's stream."""
        L.require_gpu(means, quats, log_scales, logits, sh, sky, viewmat, *grids)
        lib, st = L.lib(), L.stream()
        dev = means.device
        W, H = cfg["width"], cfg["height"]
        N = means.shape[0]
        sky = sky.contiguous()
        grids = [g.contiguous() for g in grids]

        def before_wait():   # everything of the rest of the forward that does not depend on the list counts
            idx = cfg.get("img_idx")
            sel = grids if idx is None else [g[idx:idx + 1] for g in grids]   # views: the level struct takes their addresses
            lv = _levels_struct(sel, None, cfg["factors"])
            bws_bytes = lib.bds_bilagrid_ms_workspace_bytes(len(grids), lv, H, W)
            return (_image_buffers(W, H, dev), sel, lv, bws_bytes, _empty((bws_bytes,), dev, torch.uint8), _empty((H, W, 3), dev),
                    _empty((H, W, 1), dev))

        f = _view_front(cfg, means, quats, log_scales, logits, sh, viewmat, before_wait)
        means, quats, log_scales, sh, viewmat = f.means, f.quats, f.log_scales, f.sh, f.viewmat
        scales, opac, radii, means2d, cam_pos, sh_rgb = f.scales, f.opac, f.radii, f.means2d, f.cam_pos, f.sh_rgb
        tiles_per_gauss, isect_offsets, flatten, vis_ids, M = f.tiles_per_gauss, f.isect_offsets, f.flatten, f.vis_ids, f.M
        images, sel, lv, bws_bytes, bws, rgb, depth = f.pre
        # device-count form: the gradient records (+ pose-gradient slots) exist from here on; the record pack clears them
        ctx.dev_counts = None if f.m_dev is None else (f.m_dev, f.nvis_dev)
        want_pose = bool(ctx.needs_input_grad[7])
        v_rec_all = None
        ctx.loss_rows = 0
        if f.m_dev is not None:
            # (+ the slotted accumulator of the training loss behind them: the pack's tail clear covers it, no fill launch)
            if cfg.get("train_loss") is not None and os.environ.get("BDS_FEWER_LAUNCHES", "1") == "1":
                ctx.loss_rows = L.LOSS_SLOTS * L.LOSS_SLOT_STRIDE // L.GRAD_RECORD_FLOATS
            tail_buf = cfg.get("tail_buf")     # caller-owned [POSE_GRAD_SLOTS + loss rows, 16] (graph_view: all views' slots in one tensor)
            if tail_buf is not None:
                assert tail_buf.shape[0] >= L.POSE_GRAD_SLOTS + ctx.loss_rows and tail_buf.shape[1] == L.GRAD_RECORD_FLOATS
                assert tail_buf.is_contiguous()
                v_rec_all, ctx.tail = _empty((f.n_vis, L.GRAD_RECORD_FLOATS), dev), tail_buf[:L.POSE_GRAD_SLOTS + ctx.loss_rows]
                ctx.tail_pose = tail_buf[:L.POSE_GRAD_SLOTS] if want_pose else None
            else:
                n_pose = L.POSE_GRAD_SLOTS if want_pose else 0
                v_rec_all = _empty((f.n_vis + n_pose + ctx.loss_rows, L.GRAD_RECORD_FLOATS), dev)
                ctx.tail = v_rec_all[f.n_vis:] if n_pose + ctx.loss_rows else None
                ctx.tail_pose = v_rec_all[f.n_vis:f.n_vis + n_pose] if want_pose else None
        ctx.v_rec_all = v_rec_all
        # device-count form with a backward to follow: the compositor's waves leave the schedule of their own backward (binned form:
        # no launch at all; bds_set_option(8, 0): their keys, which the sort call below orders)
        sched_buf = None
        if f.m_dev is not None and any(ctx.needs_input_grad[1:]) and _SCHEDULE_IN_FORWARD and ops._BWD_SCHEDULE:
            sp = _split(cfg, f.list_tile, f.tw * f.th, True)
            sched_buf = _empty((int(lib.bds_rasterize_schedule_ints(1, f.tw, f.th)) + int(lib.bds_rasterize_split_pool_ints(1, f.tw, f.th, sp[1], sp[2], f.M)),),
                               dev, torch.int32)
        ctx.split_ok = sched_buf is not None     # (the long-tile list of a split launch lives behind the schedule words)
        rec, render, alphas, last_ids = _composite(f, opac, images, v_rec_all, sched_buf, getattr(ctx, "tail", None))
        tiles_wh = (f.tw, f.th)
        sh_rgb, ctx.sh_by_rank = f.sh_rgb, bool(f.sh_by_rank)      # (set by the composite when the pack evaluated the colours)
        ctx.list_tile = f_list_tile = f.list_tile
        del f
        # Work of the backward that depends on the compositor's outputs only -- the longest-tile-first schedule of the compositor's
        # backward and the zeroed dense screen-space gradient arrays -- can be done HERE (_SCHEDULE_IN_FORWARD), on whichever of the
        # graph-replayed frame's two streams has the slack.
        ctx.order = None
        ctx.g2d = cfg.get("g2d_buf")       # (a caller-owned persistent buffer, kept clean row-wise by the caller: graph_view)
        if any(ctx.needs_input_grad[1:]) and _SCHEDULE_IN_FORWARD:
            if sched_buf is not None:
                L.check(lib.bds_rasterize_bwd_schedule_sort(1, tiles_wh[0], tiles_wh[1], L.ptr(sched_buf), L.stream()),
                        "bds_rasterize_bwd_schedule_sort")      # (a no-op in the binned form)
                ctx.order = sched_buf
            else:
                ctx.order = ops.bwd_schedule(1, W, H, f_list_tile, isect_offsets, last_ids)
            if ctx.g2d is None:
                ctx.g2d = torch.zeros(2, N, 2, device=dev, dtype=torch.float32)
        yield radii
        lib, st = L.lib(), L.stream()    # (the second half may be enqueued on another stream)
        # expected depth + clamp + sky blend + bilateral transform, straight from the 4-channel render
        tl = cfg.get("train_loss")     # train_view: the L1 + TV loss (value and gradients) rides on the full-resolution launch
        ctx.train_loss = None
        with L.timed("bilagrid_fwd"):
            if tl is not None:
                from .losses import loss_slots
                v_rgb_loss = _empty((H, W, 3), dev)
                if ctx.loss_rows:       # cleared by the record pack together with the gradient records
                    loss_acc = ctx.tail[ctx.tail.shape[0] - ctx.loss_rows:].view(-1)
                else:
                    loss_acc = loss_slots(dev)
                L.check(lib.bds_bilagrid_ms_ed_train_fwd(len(grids), lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(bws), bws_bytes,
                                                         L.ptr(rgb), L.ptr(depth), L.ptr(tl["target"]), len(tl["grids"]), tl["levels"],
                                                         tl["weights"], 1.0, L.ptr(loss_acc), L.LOSS_SLOTS, L.ptr(v_rgb_loss), st),
                        "bds_bilagrid_ms_ed_train_fwd")
                ctx.train_loss = (loss_acc, v_rgb_loss)     # (slotted accumulator: train_view sums it off the critical chain)
            else:
                L.check(lib.bds_bilagrid_ms_ed_fwd(len(grids), lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(bws), bws_bytes,
                                                   L.ptr(rgb), L.ptr(depth), st), "bds_bilagrid_ms_ed_fwd")
        rgb_g = render[0, :, :, :3]   # view: the Gaussians' colour before clamp / sky / transform (clamped on access, see _Out)
        # The backward's first launch (bilateral transform) is prepared HERE when its gradient targets are already known (in-place
        # grid gradients): at that point of a step the host is only tens of microseconds ahead of the GPU, and every allocation
        # or struct fill in front of that launch is GPU idle time.
        ctx.bwd_pre = None
        if cfg.get("grids_in_place") and any(ctx.needs_input_grad[8:]):
            arena_g = [cfg["grad_arena"][f"grid{i}"] for i in range(len(grids))]
            if all(a.shape == g.shape and a.is_contiguous() for a, g in zip(arena_g, grids)):
                v_g = [a if ctx.needs_input_grad[8 + i] else None for i, a in enumerate(arena_g)]
                idx = cfg.get("img_idx")
                v_sel = v_g if idx is None else [None if v is None else v[idx:idx + 1] for v in v_g]
                ctx.bwd_pre = (v_g, _levels_struct(sel, v_sel, cfg["factors"]), sel, _empty((1, H, W, 4), dev), _empty((1, H, W, 1), dev),
                               _empty((H, W, 3), dev) if ctx.needs_input_grad[6] else None)
        ctx.cfg = cfg
        ctx.M = M
        ctx.n_grids = len(grids)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(means, quats, log_scales, sh, sky, viewmat, scales, opac, radii, means2d, cam_pos, sh_rgb,
                              rec, vis_ids, flatten, isect_offsets, render, alphas, last_ids, bws, *grids)
        opacity = alphas[0]
        # rgb_g is returned for inspection.  means2d_out is a graph tensor (trainers/base.py:429-430 calls retain_grad() on
        # info["means2d"]): the backward attaches .absgrad / .grad to it, and a gradient a caller sends INTO it is added to the
        # compositor's before the projection backward.
        means2d_out = means2d.view(1, N, 2)
        ctx.mark_non_differentiable(rgb_g, radii, tiles_per_gauss, flatten, isect_offsets, vis_ids)
        return rgb, depth, opacity, rgb_g, means2d_out, radii, tiles_per_gauss, flatten, isect_offsets, vis_ids

    @staticmethod
    def backward(ctx, v_rgb, v_depth, v_opacity, _v_rgb_g, v_means2d_ext, *_):
        steps = _FusedView.backward_steps(ctx, v_rgb, v_depth, v_opacity, _v_rgb_g, v_means2d_ext)
        next(steps)                      # image half: colour transform + compositor
        try:
            next(steps)                  # Gaussian half: SH + projection over the visible rows
        except StopIteration as done:
            return done.value
        raise AssertionError("backward_steps yields once")

    @staticmethod
    def backward_steps(ctx, v_rgb, v_depth, v_opacity, _v_rgb_g, v_means2d_ext):
        """The backward as a generator that yields ONCE, between its two halves: the image half (bilateral transform, compositor:
        everything that produces the visible Gaussians' gradient records) and the Gaussian half (list-driven SH / projection
        backward into the parameter gradients, screen-space gradient arrays, pose gradient).  The second half of view v only has to
        precede the second half of view v + 1; ``graph_view`` replays it on a stream of its own next to view v + 1's image half."""
        (means, quats, log_scales, sh, sky, viewmat, scales, opac, radii, means2d, cam_pos, sh_rgb, rec, vis_ids, flatten,
         isect_offsets, render, alphas, last_ids, bws, *grids) = ctx.saved_tensors
        cfg = ctx.cfg
        lib, st = L.lib(), L.stream()
        dev = means.device
        W, H = cfg["width"], cfg["height"]
        N, K = means.shape[0], sh.shape[1]
        M, n_vis = ctx.M, vis_ids.numel()
        tw, th = math.ceil(W / TILE), math.ceil(H / TILE)
        # colour transform
        need_g = ctx.needs_input_grad[8:]
        pre = getattr(ctx, "bwd_pre", None)
        # grid gradients: one zero fill for all levels; with img_idx the full [n_img, ...] gradient is returned with only that
        # image's slice written (no slice-backward / scatter in the autograd graph)
        # ... or, with grad_arena["grid<i>"] and arena_rows >= 1, ADDED in place to the caller's accumulators (their .grad)
        arena_g = [(cfg.get("grad_arena") or {}).get(f"grid{i}") for i in range(len(grids))]
        grids_in_place = pre is not None or (bool(cfg.get("grids_in_place")) and any(need_g)
                                            and all(a is not None and a.shape == g.shape and a.is_contiguous() for a, g in zip(arena_g, grids)))
        if pre is not None:
            v_grids = pre[0]
        elif grids_in_place:
            v_grids = [a if need_g[i] else None for i, a in enumerate(arena_g)]
        else:
            sizes = [(g.numel() + 3) // 4 * 4 if need_g[i] else 0 for i, g in enumerate(grids)]
            flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32) if sum(sizes) else None
            v_grids, off = [], 0
            for i, g in enumerate(grids):
                v_grids.append(flat[off:off + g.numel()].view(g.shape) if need_g[i] else None)
                off += sizes[i]
        if pre is not None:
            _, lv, _sel_keepalive, v_render, v_alphas, v_sky = pre
            ctx.bwd_pre = pre = None   # (a second owner of v_sky would make autograd copy it instead of adopting it as sky.grad)
        else:
            idx = cfg.get("img_idx")
            sel = list(grids) if idx is None else [g[idx:idx + 1] for g in grids]
            v_sel = v_grids if idx is None else [None if v is None else v[idx:idx + 1] for v in v_grids]
            lv = _levels_struct(sel, v_sel, cfg["factors"])
            v_render, v_alphas = _empty((1, H, W, 4), dev), _empty((1, H, W, 1), dev)
            v_sky = _empty((H, W, 3), dev) if ctx.needs_input_grad[6] else None   # the sky colour is often a constant input
        v_rgb = torch.zeros(H, W, 3, device=dev) if v_rgb is None else v_rgb.contiguous()
        v_depth = None if v_depth is None else v_depth.contiguous()
        v_opacity = None if v_opacity is None else v_opacity.contiguous()
        # The transform's backward leaves its last stage -- guidance route, clamp / sky blend / expected-depth backward: a per-pixel
        # function -- to the compositor's backward wherever the configuration allows it (include/bds.h bds_rasterize_bwd_ms): v_render
        # then holds the direct-route gradient only, v_alphas is not touched, and one launch over the image is gone.
        # (a split launch -- long tiles strip by strip -- keeps those tiles out of the schedule: only bds_rasterize_bwd_dev finds them)
        split = _split(cfg, ctx.list_tile, tw * th, getattr(ctx, "split_ok", False)) if getattr(ctx, "dev_counts", None) is not None else (0, 0, 0)
        defer = (_DEFER_EPILOGUE and cfg.get("defer_epilogue", True) and M > 0 and n_vis > 0 and split[0] == 0
                 and bool(lib.bds_bilagrid_ms_ed_bwd_deferrable(len(grids), lv, H, W)))
        with L.timed("bilagrid_bwd"):
            if defer:
                L.check(lib.bds_bilagrid_ms_ed_bwd_deferred(len(grids), lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(bws),
                                                            bws.numel(), L.ptr(v_rgb), L.ptr(v_render), st), "bds_bilagrid_ms_ed_bwd_deferred")
            else:
                L.check(lib.bds_bilagrid_ms_ed_bwd(len(grids), lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(bws), bws.numel(),
                                                   L.ptr(v_rgb), L.ptr(v_depth), L.ptr(v_opacity), L.ptr(v_render), L.ptr(v_alphas),
                                                   L.ptr(v_sky), st), "bds_bilagrid_ms_ed_bwd")
        # compositing: gradient records of the visible Gaussians, in the order of vis_ids (64 bytes each)
        # (+ the camera-pose gradient slots of the projection backward behind them: one zero fill for both)
        want_pose = bool(ctx.needs_input_grad[7])
        dev_counts = getattr(ctx, "dev_counts", None)   # (M effective, visible effective) device addresses: the device-count form
        if dev_counts is not None:
            v_rec_all, ctx.v_rec_all = ctx.v_rec_all, None   # cleared by the forward's record pack
            assert v_rec_all is not None, "the device-count form runs its backward once"
            assert v_means2d_ext is None, "device-count form: no gradient into info['means2d'] (its visible-id list has no host length)"
        else:
            v_rec_all = torch.zeros(max(n_vis, 1) + (L.POSE_GRAD_SLOTS if want_pose else 0), L.GRAD_RECORD_FLOATS, device=dev,
                                    dtype=torch.float32)
        v_rec = v_rec_all[:max(n_vis, 1)]
        LT = ctx.list_tile
        order = getattr(ctx, "order", None)
        if order is None:
            order = ops.bwd_schedule(1, W, H, LT, isect_offsets, last_ids)
        with L.timed("rasterize_bwd"):
            if defer:
                L.check(lib.bds_rasterize_bwd_ms(n_vis, M, None if dev_counts is None else dev_counts[0], L.ptr(rec), W, H, TILE, LT, tw, th,
                                                 L.ptr(isect_offsets), L.ptr(flatten), L.ptr(alphas), _tfinal_ptr(alphas), L.ptr(last_ids), L.ptr(v_rec), 1,
                                                 L.ptr(order), len(grids), lv, L.ptr(bws), bws.numel(), L.ptr(render), L.ptr(sky),
                                                 L.ptr(v_depth), L.ptr(v_opacity), L.ptr(v_render), L.ptr(v_sky), st), "bds_rasterize_bwd_ms")
            elif dev_counts is not None:
                L.check(lib.bds_rasterize_bwd_dev(1, n_vis, M, dev_counts[0], 4, L.ptr(rec), None, W, H, TILE, LT, tw, th, L.ptr(isect_offsets),
                                                  L.ptr(flatten), L.ptr(alphas), _tfinal_ptr(alphas), L.ptr(last_ids), L.ptr(v_render),
                                                  L.ptr(v_alphas), L.ptr(v_rec), 1, L.ptr(order), *split, st), "bds_rasterize_bwd_dev")
            else:
                L.check(lib.bds_rasterize_bwd(1, n_vis, M, 4, L.ptr(rec), None, W, H, TILE, LT, tw, th, L.ptr(isect_offsets), L.ptr(flatten),
                                              L.ptr(alphas), _tfinal_ptr(alphas), L.ptr(last_ids), L.ptr(v_render), L.ptr(v_alphas), L.ptr(v_rec), 1,
                                              L.ptr(order), st), "bds_rasterize_bwd")
        if v_means2d_ext is not None and n_vis:   # a loss term on info["means2d"] itself: add its rows to the records
            v_rec[:n_vis, 7:9] += v_means2d_ext.reshape(N, 2).index_select(0, vis_ids.long())
        yield
        lib, st = L.lib(), L.stream()    # (the second half may be enqueued on another stream)
        # dense screen-space gradient + its absolute sum for the densification statistics (zeros for culled Gaussians, as gsplat)
        g2d, ctx.g2d = getattr(ctx, "g2d", None), None
        if g2d is None:
            g2d = torch.zeros(2, N, 2, device=dev, dtype=torch.float32)
        arena = cfg.get("grad_arena") or {}
        # arena modes (need all five per-Gaussian arena entries): 1 = store the visible rows into an arena the caller keeps zero
        # elsewhere, 2 = add them to an arena that already is the parameters' .grad (several views summed before one exchange)
        rows = int(cfg.get("arena_rows", 0)) if all(k in arena for k in ("means", "quats", "log_scales", "opacity_logits", "sh")) else 0
        # sink mode (multi-GPU, dist.FrameExchange): the rows go into a compact exchange buffer, row_map[g] = the slot of Gaussian g
        # in the union of the ranks' visible sets; the sink adds the reduced rows to the parameters' .grad itself
        sink, row_map = cfg.get("grad_sink"), None
        if sink is not None:
            arena, row_map = sink.targets(vis_ids)
            rows = 1

        def out_like(name, ref):  # gradient output: the caller's slice of a flat communication buffer, or a fresh zero tensor
            t = arena.get(name)
            if t is None:
                return torch.zeros_like(ref)
            # (contiguous, or a column of the gradients' [N,16] row block: dist.FlatGradients(row_block=True))
            assert t.dtype == ref.dtype and t.device == ref.device and t.shape[1:] == ref.shape[1:], name
            assert t.is_contiguous() or (t.stride(0) == 16 and (t.dim() == 1 or t.stride(-1) == 1)), name
            assert sink is not None or t.shape == ref.shape, name
            if not rows:
                t.zero_()
            return t.view(t.shape)  # a fresh tensor object (no other owner): autograd adopts it as .grad without cloning

        v_sh = out_like("sh", sh)
        with L.timed("sh_bwd"):
            st = L.stream()
            if dev_counts is not None:
                L.check(lib.bds_sh_view_bwd_list_dev(n_vis, dev_counts[1], L.ptr(vis_ids), K, cfg["sh_degree"], L.ptr(means), L.ptr(cam_pos),
                                                     L.ptr(sh_rgb), int(bool(getattr(ctx, "sh_by_rank", False))), L.ptr(v_rec), L.ptr(v_sh),
                                                     L.ptr(row_map), int(rows == 2), st),
                        "bds_sh_view_bwd_list_dev")
            else:
                L.check(lib.bds_sh_view_bwd_list(n_vis, L.ptr(vis_ids), K, cfg["sh_degree"], L.ptr(means), L.ptr(cam_pos), L.ptr(sh_rgb),
                                                 int(bool(getattr(ctx, "sh_by_rank", False))), L.ptr(v_rec), L.ptr(v_sh), L.ptr(row_map),
                                                 int(rows == 2), st), "bds_sh_view_bwd_list")
        st = L.stream()
        v_means, v_quats = out_like("means", means), out_like("quats", quats)
        v_ls, v_logits = out_like("log_scales", log_scales), out_like("opacity_logits", opac)
        Kmat = cfg["K"].contiguous()
        if dev_counts is not None:
            tp = getattr(ctx, "tail_pose", None)
            v_vm_slots = tp.view(L.POSE_GRAD_SLOTS, 4, 4) if (want_pose and tp is not None) else None
        else:
            v_vm_slots = (v_rec_all[max(n_vis, 1):max(n_vis, 1) + L.POSE_GRAD_SLOTS].view(L.POSE_GRAD_SLOTS, 4, 4)
                          if want_pose else None)   # camera-pose gradient (base.py:328-329,399)
        with L.timed("project_bwd"):
            if dev_counts is not None:
                L.check(lib.bds_project_view_bwd_list_dev(n_vis, dev_counts[1], L.ptr(vis_ids), L.ptr(means), L.ptr(quats), L.ptr(scales),
                                                          L.ptr(opac), L.ptr(viewmat.contiguous()), L.ptr(Kmat), W, H, cfg["eps2d"], L.ptr(v_rec),
                                                          _dp(v_means), _dp(v_quats), _dp(v_ls), _dp(v_logits), L.ptr(v_vm_slots),
                                                          L.ptr(g2d[0]), L.ptr(g2d[1]), L.ptr(row_map), int(rows == 2), st),
                        "bds_project_view_bwd_list_dev")
            else:
                L.check(lib.bds_project_view_bwd_list(n_vis, L.ptr(vis_ids), L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opac),
                                                      L.ptr(viewmat.contiguous()), L.ptr(Kmat), W, H, cfg["eps2d"], L.ptr(v_rec), _dp(v_means),
                                                      _dp(v_quats), _dp(v_ls), _dp(v_logits), L.ptr(v_vm_slots), L.ptr(g2d[0]), L.ptr(g2d[1]),
                                                      L.ptr(row_map), int(rows == 2), st), "bds_project_view_bwd_list")
        carrier = cfg["_means2d_ref"]() if cfg.get("_means2d_ref") is not None else None
        if carrier is not None:  # the tensor the caller holds in info["means2d"] (trainers/base.py:282-284 read .absgrad / .grad)
            carrier.grad = g2d[0:1]
            carrier.absgrad = g2d[1:2]
        # (defer_pose_sum: the caller sums the slots of all its views in one launch -- graph_view, once per frame)
        v_viewmat = None if (v_vm_slots is None or cfg.get("defer_pose_sum")) else v_vm_slots.sum(0)
        if grids_in_place:
            v_grids = [None] * len(grids)
        if rows == 2 or sink is not None:   # already added in place to what autograd holds as .grad (or handed to the sink)
            return (None, None, None, None, None, None, v_sky, v_viewmat, *v_grids)
        return (None, v_means, v_quats, v_ls, v_logits, v_sh, v_sky, v_viewmat, *v_grids)


def fused_view(params: Dict[str, Tensor], viewmat: Tensor, K: Tensor, width: int, height: int, grids: Sequence[Tensor],
               sky: Tensor, factors: Sequence[int], sh_degree: int = 3, near_plane: float = 0.1, far_plane: float = 1e10,
               radius_clip: float = 0.0, eps2d: float = 0.3, tile_cull: bool = True,
               grad_arena: Optional[Dict[str, Tensor]] = None, cam_pos: Optional[Tensor] = None,
               img_idx: Optional[int] = None, arena_rows: int = 0, grad_sink=None, list_tile: Optional[int] = None,
               caps: Optional[ListCapacity] = None, prep_ws: Optional[Tensor] = None):
    """params: means [N,3], quats [N,4] (raw), log_scales [N,3], opacity_logits [N], sh [N,16,3];
    grids: per level [1,12,L,gy,gx] (the current image's grids), or -- with ``img_idx`` -- the full parameters
    [n_img,12,L,gy,gx] of which image ``img_idx`` is used (models/modules.py:507-512); the gradient then comes back in the
    parameter's shape.  ``viewmat`` [4,4] may require grad (learnable camera pose, trainers/base.py:328-329,399): its gradient
    comes from the projection; the SH view direction uses the detached camera centre as the reference does (vanilla.py:385 ``.data``).
    Returns dict(rgb, depth, opacity, rgb_gaussians, info); ``info["means2d"]`` [1,N,2] is a graph tensor that supports
    ``retain_grad()`` and carries ``.absgrad`` / ``.grad`` after ``backward()`` (trainers/base.py:279-297,429-430).

    ``grad_arena`` (optional): name -> preallocated tensor; the backward kernels write the parameter gradients
    straight into these (e.g. slices of the flat all-reduce buffer of ``dist.FlatGradients``) instead of fresh
    tensors, so that multi-GPU runs need no pack pass.  Valid when each parameter receives its gradient from
    this node only (the reference's step: one view per iteration).
    ``arena_rows`` (opt-in, needs all five per-Gaussian arena entries): 1 = the backward STORES only the rows of the Gaussians this view
    sees -- the caller promises that every other row of the arena is zero (``dist.FlatGradients(sparse_rows=True).zero()`` keeps it so);
    2 = it ADDS them to an arena that autograd already holds as the parameters' ``.grad`` (second and later views of a frame that
    is exchanged once) and returns no gradient for those parameters.
    ``grad_sink`` (multi-GPU, ``dist.FrameExchange``): an object whose ``targets(visible_ids)`` is called inside the backward and
    returns (compact buffers by name, row_map [N] i32); the visible rows are stored at ``row_map[g]`` of those buffers and no
    gradient is returned for the five per-Gaussian parameters (the sink adds the reduced rows to their ``.grad``).
    ``caps`` (``ListCapacity``): the device-count form -- no host wait at all; the lists are built into buffers of these capacities
    and every kernel takes its counts from device memory (what ``graph_view.ViewGraph`` captures in a hipGraph).  ``info["n_isects"]``
    / ``["n_visible"]`` are then the CAPACITIES (the lists' tensors have those lengths; entries beyond the counts are undefined);
    the actual counts arrive in ``caps.counts``."""
    if cam_pos is None:  # camera centre (vanilla.py:385 uses camtoworlds.data[..., :3, 3]); callers with fixed cameras cache it
        cam_pos = torch.linalg.inv(viewmat.detach())[:3, 3].contiguous()
    cfg = dict(width=int(width), height=int(height), K=K, cam_pos=cam_pos.detach(), factors=tuple(int(f) for f in factors),
               sh_degree=int(sh_degree), near_plane=float(near_plane), far_plane=float(far_plane), radius_clip=float(radius_clip),
               eps2d=float(eps2d), tile_cull=bool(tile_cull), grad_arena=grad_arena, grad_sink=grad_sink,
               img_idx=None if img_idx is None else int(img_idx), arena_rows=int(arena_rows),
               list_tile=int(LIST_TILE if list_tile is None else list_tile), caps=caps, prep_ws=prep_ws)
    gs = [g if g.dim() == 5 else g[None] for g in grids]
    # in-place grid gradients only when the arena entries ARE the grids' .grad right now (dist.FrameExchange.begin_frame sets that up)
    if grad_arena is not None and int(arena_rows) >= 1 and grad_sink is None:
        cfg["grids_in_place"] = all(g.grad is not None and grad_arena.get(f"grid{i}") is not None
                                    and g.grad.data_ptr() == grad_arena[f"grid{i}"].data_ptr() for i, g in enumerate(gs))
    out = _FusedView.apply(cfg, params["means"], params["quats"], params["log_scales"], params["opacity_logits"], params["sh"], sky,
                           viewmat, *gs)
    rgb, depth, opacity, rgb_g, means2d, radii, tiles_per_gauss, flatten_ranks, isect_offsets, vis_ids = out
    cfg["_means2d_ref"] = weakref.ref(means2d)  # backward attaches .absgrad to THIS tensor object
    info = _Info({"means2d": means2d, "radii": radii, "width": int(width), "height": int(height), "tiles_per_gauss": tiles_per_gauss,
                  "flatten_ranks": flatten_ranks, "visible_ids": vis_ids, "isect_offsets": isect_offsets,
                  "tile_size": cfg["list_tile"],   # of the lists in this dict (the compositor subdivides into 16 x 16)
                  "n_cameras": 1, "n_isects": int(flatten_ranks.numel()), "n_visible": int(vis_ids.numel())})
    return _Out(rgb=rgb, depth=depth, opacity=opacity, _rgb_g_raw=rgb_g, _sky=sky, info=info)


@torch.no_grad()
def render_classes(params: Dict[str, Tensor], viewmat: Tensor, K: Tensor, width: int, height: int, masks: Dict[str, Tensor],
                   sh_degree: int = 3, near_plane: float = 0.1, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
                   tile_cull: bool = True, cam_pos: Optional[Tensor] = None, include_full: bool = True,
                   list_tile: Optional[int] = None) -> Dict[str, Tensor]:
    """Evaluation re-renders of Gaussian subsets (per-class and "Dynamic" images, trainers/scene_graph.py:296-313): the reference calls
    its ``render_fn(gaussian_mask)`` once per class, i.e. the whole ``rasterization`` again with ``opacities * mask``
    (trainers/base.py:392-416).  Projection, SH colours, the tile lists and their sort do not depend on the mask, so here they are
    computed ONCE and only the splat-record pack + forward composite run per mask -- a masked-out Gaussian has opacity 0, fails the
    alpha >= 1/255 test and leaves the transmittance untouched, exactly as in the reference's re-render.

    masks: name -> [N] bool / float.  Returns ``{name+"_rgb": [H,W,3] (clamped at 1), name+"_depth": [H,W,1] expected depth,
    name+"_opacity": [H,W,1]}`` for every mask, plus ``rgb_gaussians`` / ``depth`` / ``opacity`` of the unmasked scene with
    ``include_full``."""
    if cam_pos is None:
        cam_pos = torch.linalg.inv(viewmat)[:3, 3].contiguous()
    cfg = dict(width=int(width), height=int(height), K=K, cam_pos=cam_pos, sh_degree=int(sh_degree), near_plane=float(near_plane),
               far_plane=float(far_plane), radius_clip=float(radius_clip), eps2d=float(eps2d), tile_cull=bool(tile_cull),
               list_tile=int(LIST_TILE if list_tile is None else list_tile),
               sh_in_pack=False)   # the colours are evaluated once and packed with every mask's opacities
    f = _view_front(cfg, params["means"].detach(), params["quats"].detach(), params["log_scales"].detach(),
                    params["opacity_logits"].detach(), params["sh"].detach(), viewmat.detach())

    def image(opac):
        _, render, alphas, _ = _composite(f, opac.contiguous())
        a = alphas[0]
        return torch.clamp(render[0, :, :, :3], max=1.0), render[0, :, :, 3:4] / a.clamp(min=1e-10), a

    out: Dict[str, Tensor] = {}
    if include_full:
        out["rgb_gaussians"], out["depth"], out["opacity"] = image(f.opac)
    for name, m in masks.items():
        assert m.shape == (f.N,), (name, tuple(m.shape), f.N)
        out[name + "_rgb"], out[name + "_depth"], out[name + "_opacity"] = image(f.opac * m.to(f.opac.dtype))
    return out


class _DirectCtx:
    """Stands in for the autograd context when the forward / backward bodies of a Function are called directly (train_view)."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


_ONES: Dict[torch.device, Tensor] = {}


def _accumulate(p: Tensor, g: Optional[Tensor]) -> None:
    """What autograd's AccumulateGrad does for a leaf: adopt, or add unless the producer already wrote into ``p.grad`` itself."""
    if g is None or not p.requires_grad:
        return
    if p.grad is None:
        p.grad = g
    elif p.grad.data_ptr() != g.data_ptr():
        p.grad.add_(g)


# Where the compositor backward's schedule (two launches) and the zeroed screen-space gradient arrays (one fill) are produced: behind
# the compositor's forward (1) or in front of its backward (0).  Measured on the two-stream frame: 937 vs 919 it/s -- in front of the
# compositor's backward they delay the one kernel whose end closes the phase both compositors share.
_SCHEDULE_IN_FORWARD = os.environ.get("BDS_SCHEDULE_IN_FORWARD", "1") == "1"
_PROJECT_PREPARES = os.environ.get("BDS_PROJECT_PREPARES", "1") == "1"   # (A/B: 0 = projection and visible-reduce as two launches)
_LOSS_TWO_STEP = os.environ.get("BDS_LOSS_TWO_STEP", "0") == "1"   # ablation: the loss as forward + backward launches
_LOSS_IN_TRANSFORM = os.environ.get("BDS_LOSS_IN_TRANSFORM", "1") == "1"   # the loss rides on the colour transform's launch


def train_view(params: Dict[str, Tensor], viewmat: Tensor, K: Tensor, width: int, height: int, grids: Sequence[Tensor], sky: Tensor,
               factors: Sequence[int], target: Tensor, tv_weights: Sequence[float], img_idx: Optional[int] = None,
               grid_grads: Optional[Sequence[Tensor]] = None, after_forward=None, **kwargs):
    """One training view WITHOUT an autograd graph: the forward of ``fused_view``, the L1 + per-level TV loss of
    ``losses.photometric_tv_loss`` and both backward passes run back to back on the calling thread -- the same kernels in the same
    order as ``loss = photometric_tv_loss(fused_view(...)["rgb"], target, grids, tv_weights); loss.backward()``, minus the hand-over
    to autograd's device thread (~50 us of idle GPU per view at 1080p) and its bookkeeping.  Gradients land where autograd would put
    them: per-Gaussian rows in ``grad_arena`` (``arena_rows``) or in ``param.grad``; the grids' in ``grid_grads`` / ``.grad``;
    ``sky.grad``, ``viewmat.grad`` accumulated.  Accepts the keyword arguments of ``fused_view``; ``after_forward(info)`` is called
    between the forward and the backward pass (``dist.FrameExchange.begin_view`` starts its visibility exchange there).
    ``two_phase=True``: only the forward, the loss value and the loss gradient are enqueued; the returned dict carries ``backward``
    and ``backward_tail``, callables that enqueue the rest (once each, in this order: see ``_FusedView.backward_steps``); the
    ``loss`` entry may appear as late as ``backward_tail`` (its value is summed from the loss launch's slotted accumulator there, off
    the critical chain).  Returns dict(loss, rgb, depth, opacity, info): detached tensors."""
    from .losses import _PhotometricTV, photometric_tv_train, slots_value
    cam_pos = kwargs.pop("cam_pos", None)
    if cam_pos is None:
        cam_pos = torch.linalg.inv(viewmat.detach())[:3, 3].contiguous()
    grad_arena, arena_rows = kwargs.pop("grad_arena", None), int(kwargs.pop("arena_rows", 0))
    grad_sink = kwargs.pop("grad_sink", None)
    list_tile, caps = kwargs.pop("list_tile", None), kwargs.pop("caps", None)
    prep_ws, two_phase = kwargs.pop("prep_ws", None), bool(kwargs.pop("two_phase", False))
    g2d_buf = kwargs.pop("g2d_buf", None)    # persistent [2,N,2] screen-space gradient arrays whose stale rows the caller clears
    tail_buf, defer_pose_sum = kwargs.pop("tail_buf", None), bool(kwargs.pop("defer_pose_sum", False))
    lazy_loss = bool(kwargs.pop("lazy_loss", False))    # leave the loss value as out["loss_slots"] (losses.slots_value sums it on demand)
    defer_epilogue = bool(kwargs.pop("defer_epilogue", True))   # (see _DEFER_EPILOGUE)
    split_len, split_cap = kwargs.pop("split_len", None), kwargs.pop("split_cap", None)   # device-count compositors: long tiles strip by strip (_split)
    block_bounds = kwargs.pop("block_bounds", None)    # bds_gaussian_block_bounds of the current parameters (the projection skips whole blocks)
    row_catchup = kwargs.pop("row_catchup", None)      # optim.DeferredRowAdam.catchup (device-count form with the SH colours in the pack)
    # the TV term over OTHER tensors than the transform's grids: graph_view's replayable view slices staging copies of ONE image's
    # grids (picked by a device-side index) while the regulariser runs over the full [n_img, ...] parameters (modules.py:445)
    tv_grids, tv_grid_grads = kwargs.pop("tv_grids", None), kwargs.pop("tv_grid_grads", None)
    opts = dict(sh_degree=3, near_plane=0.1, far_plane=1e10, radius_clip=0.0, eps2d=0.3, tile_cull=True)
    opts.update({k: kwargs.pop(k) for k in list(kwargs) if k in opts})
    assert not kwargs, f"unknown arguments {sorted(kwargs)}"
    cfg = dict(width=int(width), height=int(height), K=K, cam_pos=cam_pos.detach(), factors=tuple(int(f) for f in factors),
               sh_degree=int(opts["sh_degree"]), near_plane=float(opts["near_plane"]), far_plane=float(opts["far_plane"]),
               radius_clip=float(opts["radius_clip"]), eps2d=float(opts["eps2d"]), tile_cull=bool(opts["tile_cull"]),
               grad_arena=grad_arena, grad_sink=grad_sink, img_idx=None if img_idx is None else int(img_idx), arena_rows=arena_rows,
               list_tile=int(LIST_TILE if list_tile is None else list_tile), caps=caps, prep_ws=prep_ws, g2d_buf=g2d_buf, tail_buf=tail_buf,
               defer_epilogue=defer_epilogue, split_len=split_len, split_cap=split_cap, block_bounds=block_bounds,
               defer_pose_sum=defer_pose_sum, row_catchup=row_catchup)
    gs = [g if g.dim() == 5 else g[None] for g in grids]
    if grad_arena is not None and (arena_rows >= 1 or grad_sink is not None):
        # (with a sink the arena names the GRID gradients only: the per-Gaussian rows go to the sink's compact buffers)
        cfg["grids_in_place"] = all(g.grad is not None and grad_arena.get(f"grid{i}") is not None
                                    and g.grad.data_ptr() == grad_arena[f"grid{i}"].data_ptr() for i, g in enumerate(gs))
    if (grid_grads is not None and _LOSS_IN_TRANSFORM and not _LOSS_TWO_STEP and all(g.requires_grad and g.is_contiguous() for g in gs)
            and all(a.is_contiguous() and a.numel() == g.numel() for a, g in zip(grid_grads, gs)) and target.is_contiguous()
            and target.dtype == torch.float32 and tuple(target.shape) == (int(height), int(width), 3) and target.data_ptr() % 16 == 0):
        from .losses import _levels_struct as _tv_levels
        tvg = gs if tv_grids is None else [g if g.dim() == 5 else g[None] for g in tv_grids]
        tvgg = list(grid_grads) if tv_grid_grads is None else list(tv_grid_grads)
        cfg["train_loss"] = dict(target=target, grids=tvg, grid_grads=tvgg, levels=_tv_levels(tvg, tvgg, [1] * len(tvg)),
                                 weights=(C.c_float * max(len(tvg), 1))(*[float(w) for w in tv_weights]))
    names = ("means", "quats", "log_scales", "opacity_logits", "sh")
    leaves = [params[k] for k in names]
    needs = (False, *[bool(t.requires_grad) for t in leaves], bool(sky.requires_grad), bool(viewmat.requires_grad),
             *[bool(g.requires_grad) for g in gs])
    ctx = _DirectCtx(needs)
    state = {}
    out = _Out()

    def image_half(fsteps):
        """Second half of the forward (colour transform), the L1 + TV loss and its backward with d(loss) = 1 (the TV term's gradient
        is added to the grids' gradient slices with atomics, so it may run next to another view's backward)."""
        try:
            next(fsteps)
            raise AssertionError("forward_steps yields once")
        except StopIteration as done:
            (rgb, depth, opacity, rgb_g, means2d, radii, tiles_per_gauss, flatten_ranks, isect_offsets, vis_ids) = done.value
        cfg["_means2d_ref"] = weakref.ref(means2d)
        info = _Info({"means2d": means2d, "radii": radii, "width": int(width), "height": int(height), "tiles_per_gauss": tiles_per_gauss,
                      "flatten_ranks": flatten_ranks, "visible_ids": vis_ids, "isect_offsets": isect_offsets,
                      "tile_size": cfg["list_tile"], "n_cameras": 1, "n_isects": int(flatten_ranks.numel()),
                      "n_visible": int(vis_ids.numel())})
        if after_forward is not None:
            after_forward(info)
        gg = None if grid_grads is None else list(grid_grads)
        loss = None
        if ctx.train_loss is not None:      # computed by the colour transform's own launch (forward_steps)
            (state["loss_slots"], state["v_rgb"]), ctx.train_loss = ctx.train_loss, None
            state["v_tv_grids"] = (None,) * len(gs)
        elif gg is not None and all(g.requires_grad for g in gs) and all(a.is_contiguous() for a in gg) and not _LOSS_TWO_STEP:
            # value and gradient in one launch (d(loss) = 1 is known up front); the TV gradient goes straight to the grids' slices
            state["loss_slots"], state["v_rgb"] = photometric_tv_train(rgb, target, gs, tv_weights, gg, slots=True)
            state["v_tv_grids"] = (None,) * len(gs)
        else:
            lctx = _DirectCtx((True, False, False, False, *[bool(g.requires_grad) for g in gs]))
            loss = _PhotometricTV.forward(lctx, rgb, target, tuple(float(w) for w in tv_weights), gg, *gs)
            one = _ONES.get(rgb.device)
            if one is None:
                one = _ONES[rgb.device] = torch.ones((), device=rgb.device, dtype=torch.float32)
            lg = _PhotometricTV.backward(lctx, one)
            state["v_rgb"], state["v_tv_grids"] = lg[0], lg[4:]
        out.update(loss=loss, rgb=rgb, depth=depth, opacity=opacity, _rgb_g_raw=rgb_g, _sky=sky, info=info)

    with torch.no_grad():
        fsteps = _FusedView.forward_steps(ctx, cfg, *leaves, sky, viewmat, *gs)
        out["radii"] = next(fsteps)          # (available before the colour transform: the exchange's visibility mask)
        image_half(fsteps)

    def backward():          # image half of the view's backward
        with torch.no_grad():
            state["steps"] = _FusedView.backward_steps(ctx, state["v_rgb"], None, None, None, None)
            next(state["steps"])

    def backward_tail():     # Gaussian half; gradients land where autograd would put them
        with torch.no_grad():
            if state.get("loss_slots") is not None:      # the loss VALUE: a small reduction nobody waits for
                out["loss_slots"] = state.pop("loss_slots")
                if not lazy_loss:
                    out["loss"] = slots_value(out["loss_slots"])
            try:
                next(state["steps"])
                raise AssertionError("backward_steps yields once")
            except StopIteration as done:
                grads = done.value
            for p, g in zip(leaves, grads[1:6]):
                _accumulate(p, g)
            _accumulate(sky, grads[6])
            _accumulate(viewmat, grads[7])
            for g, a, b in zip(gs, state["v_tv_grids"], grads[8:]):
                _accumulate(g, a)
                _accumulate(g, b)

    if two_phase:     # the caller enqueues the halves itself (graph_view: forward | backward [| its Gaussian half] as hipGraphs)
        out["backward"], out["backward_tail"] = backward, backward_tail
        return out
    backward()
    backward_tail()
    return out
