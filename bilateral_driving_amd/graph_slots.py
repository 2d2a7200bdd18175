"""View slots, list capacities and the validity protocol of ``graph_view.FrameGraph`` (a mixin: the capture / replay logic lives there).

* ``ViewGraph``: one captured view (forward graph | backward graph [| Gaussian-half graph]) and its static outputs.
* replayable slots (``dynamic=True``): ``set_view`` rewrites a slot's static inputs, the slot's image's bilateral grids are picked on
  the device (bds_bilagrid_select; an out-of-range device-side index raises in ``valid()``) -- the reference's loop draws a random
  image per step (/root/reference/project/tools/train.py:250-283).
* capacities: ``calibrate`` sizes the device-count lists (and, with an exchange, the ranks' unions) from a host-count visit of the
  calibration cameras; a list that outgrows its capacity renders nothing and is seen by ``valid()`` after the fact, which grows the
  capacities (never shrinks; dynamic: every slot together) and captures again -- with an exchange as a COLLECTIVE decision.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import _lib as L
from . import harness as Hn
from .fused_view import ListCapacity


class ViewGraph:
    """One captured view: ``graph`` = forward + loss value, ``graph_bwd`` = backward (with an exchange: its image half, and
    ``graph_tail`` the Gaussian half, the collectives between them)."""

    def __init__(self, graph, out, caps: ListCapacity, prep_ws: Tensor, sky: Tensor, viewmat: Tensor, graph_bwd, graph_tail=None):
        self.graph, self.graph_bwd, self.graph_tail, self.out, self.caps, self.prep_ws = graph, graph_bwd, graph_tail, out, caps, prep_ws
        self._loss_slots = out.get("loss_slots")
        self.rgb, self.depth, self.opacity = out["rgb"], out["depth"], out["opacity"]
        self.v_sky, self.v_viewmat = sky.grad, viewmat.grad
        self.done, self.fwd_done = torch.cuda.Event(), torch.cuda.Event()

    @property
    def loss(self) -> Tensor:
        """The view's loss value (read it before the next ``step()``: the next replay's record pack clears the accumulator)."""
        from .losses import slots_value
        torch.cuda.current_stream(self._loss_slots.device).wait_event(self.done)
        return slots_value(self._loss_slots)


def camera_centre(viewmat: Tensor) -> Tensor:
    """Camera centre in world space of a rigid world -> camera matrix, -R^T t (no host wait, unlike a general inverse)."""
    vm = viewmat.detach()
    return -(vm[:3, :3].transpose(0, 1) @ vm[:3, 3]).contiguous()


BLOCK_BOUNDS = __import__("os").environ.get("BDS_BLOCK_BOUNDS", "1") == "1"   # the projection skips 256-row blocks no centre of which can be visible (include/bds.h bds_gaussian_block_bounds)
GRAD_ROWS = __import__("os").environ.get("BDS_GRAD_ROWS", "1") == "1"   # the four small per-Gaussian gradients as one [N,16] row block (dist.FlatGradients(row_block=True); 0 = one array each)
SPLIT_LIST_LEN = int(__import__("os").environ.get("BDS_SPLIT_LIST_LEN", "6144"))   # entries of a list-tile list from which its tiles go strip by strip


def split_len_for(info):
    """(split_len, split_cap) for a camera from the lists of a host-count visit ((0, 0) = off): one wave per tile runs as long as the
    tile's list, so a view in which a few list tiles hold several times the entries of a typical busy one (the vanishing point of a
    street full of small splats) waits for those waves; their tiles are then composited by four waves (include/bds.h
    bds_rasterize_fwd_dev).  Off unless some list reaches ``SPLIT_LIST_LEN`` entries AND four times the mean list: a view of evenly
    long lists gains nothing from more workgroups.  Capacity: twice the 16-px tiles of the lists that long now, + 64."""
    if SPLIT_LIST_LEN <= 0:
        return 0, 0
    offs = info["isect_offsets"].reshape(-1).long()
    if offs.numel() < 2:
        return 0, 0
    M = int(info["n_isects"])
    lens = torch.diff(offs, append=offs.new_tensor([M]))
    longest, mean = int(lens.max()), float(lens.float().mean())
    if not (longest >= SPLIT_LIST_LEN and longest >= 4.0 * mean):
        return 0, 0
    div = max(int(info.get("tile_size", 64)) // 16, 1)
    return SPLIT_LIST_LEN, 2 * int((lens >= SPLIT_LIST_LEN).sum()) * div * div + 64


class FrameCapacities:
    # ---- replayable views ------------------------------------------------------------------------------------------------------
    def set_view(self, v: int, cam: Hn.Camera, target: Tensor, sky: Tensor, img_idx) -> None:
        """Slot v renders ``cam`` against ``target`` with image ``img_idx``'s grids from the next ``step()`` on (``dynamic=True``).
        Copies into the slot's static inputs on the current stream -- no host wait; ``img_idx``: int or a device int tensor."""
        assert self.dynamic, "FrameGraph(dynamic=True) owns rewritable view slots"
        slot = self.cams[v]
        assert cam.width == slot.width and cam.height == slot.height, "one FrameGraph per image size"
        with torch.no_grad():
            slot.viewmat.copy_(cam.viewmat, non_blocking=True)
            slot.K.copy_(cam.K, non_blocking=True)
            slot.cam_pos.copy_(cam.cam_pos if cam.cam_pos is not None else camera_centre(slot.viewmat), non_blocking=True)
            self.targets[v].copy_(target, non_blocking=True)
            self.skies[v].copy_(sky, non_blocking=True)
            if isinstance(img_idx, Tensor):      # (a device-side index: an out-of-range value is reported by valid())
                self.img_idx_dev[v].copy_(img_idx.reshape(1), non_blocking=True)
            else:
                assert 0 <= int(img_idx) < self.grids[0].shape[0], f"image index {img_idx} outside the {self.grids[0].shape[0]} grids"
                self.img_idx_dev[v].fill_(int(img_idx))
                self.img_indices[v] = int(img_idx)

    def _select_levels(self, bwd: bool):
        from .bilagrid import _levels_struct
        return _levels_struct(self.grids, [self.arena[f"grid{i}"] for i in range(len(self.grids))] if bwd else None, [1] * len(self.grids))

    def _grid_select(self, v: int, bwd: bool) -> None:
        """dynamic: the slot's image's grids into its staging copies (forward) / its staging gradients into the parameters' gradient
        slices, and cleared (backward) -- one launch each, the image index read on the device."""
        bufs = self.sel_grads[v] if bwd else self.sel_grids[v]
        ptrs = (C.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
        fn = L.lib().bds_bilagrid_select_bwd if bwd else L.lib().bds_bilagrid_select
        L.check(fn(len(bufs), self._select_levels(bwd), L.ptr(self.img_idx_dev[v]), ptrs, self.sel_err[v].data_ptr(), L.stream()),
                "bds_bilagrid_select")

    # ---- capacities --------------------------------------------------------------------------------------------------------------
    def calibrate(self) -> None:
        """One forward visit of every calibration camera through the host-count path: the list capacities (and, with an exchange, the
        size of the ranks' union per view) are sized from what it needed -- per view, or (dynamic) the maximum for every slot."""
        import torch.distributed as dist
        with torch.no_grad():
            if self.dynamic:
                M = n_vis = 0
                split = (0, 0)
                for cam in self.calib_cams:
                    info = Hn.render_view(self.params, cam, self.grids, 0, self.skies[0], factors=self.factors, sh_degree=self.sh_degree,
                                          list_tile=self.list_tile)["info"]
                    M, n_vis = max(M, int(info["n_isects"])), max(n_vis, int(info["n_visible"]))
                    split = max(split, split_len_for(info), key=lambda t: t[1])
                for v in range(self.V):
                    self._grow(v, M, n_vis)
                    self.split_len[v], self.split_cap[v] = split
                if self.fx is not None:
                    # any camera of any rank's sweep may meet any camera of another's: the union of a slot over the ranks is at most
                    # the SUM of the ranks' largest visible sets (and at most the scene) -- the same number on every rank
                    from .dist import dynamic_union_bound
                    self._unions = [dynamic_union_bound(n_vis, self.N, self.dev if self.fx.coll else None)] * self.V
                return
            for v, cam in enumerate(self.cams):
                info = Hn.render_view(self.params, cam, self.grids, self.img_indices[v], self.skies[v], factors=self.factors,
                                      sh_degree=self.sh_degree, list_tile=self.list_tile)["info"]
                self._grow(v, int(info["n_isects"]), int(info["n_visible"]))
                self.split_len[v], self.split_cap[v] = split_len_for(info)
                if self.fx is not None:
                    mask = (info["radii"].reshape(-1) > 0).to(torch.uint8)
                    if self.fx.coll:
                        from .dist import _all_reduce
                        _all_reduce(mask, dist.ReduceOp.MAX)
                    self._unions[v] = max(self._unions[v], int(mask.sum()))

    def _grow(self, v: int, M: int, n_vis: int) -> None:
        old = self.caps[v]
        h = self.headroom if old is None else max(self.headroom, 1.25)   # (a re-provision is a re-capture: make it worth it)
        m_cap = max(int(M * h) + 4096, old.m_cap if old else 0)
        nv_cap = max(min(int(n_vis * h) + 1024, self.N), old.nvis_cap if old else 0)   # never shrinks
        self.caps[v] = ListCapacity(m_cap, max(nv_cap, 1))

    # ---- after the fact ----------------------------------------------------------------------------------------------------------
    def counts(self):
        """[(M, visible)] per view as last written by the GPU (synchronise first for the current frame's values)."""
        return [c.observed() for c in self.caps]

    def row_lists(self):
        """[(capacity, device address of the count, device address of the int32 id list)] of the views' visible Gaussians as the last
        ``step()`` left them in the prepare workspaces -- what a row-lazy optimizer steps over (``optim.DeferredRowAdam.step``)."""
        return [(self.caps[v].nvis_cap, self.prep_ws[v].data_ptr() + self._nvis_off, self.prep_ws[v].data_ptr() + self._ids_off)
                for v in range(self.V)]

    def _check_counts(self, raise_on_overflow: bool = False) -> bool:
        ok = True
        seen = [c.wanted() for c in self.caps]
        for v, c in enumerate(self.caps):
            M, n_vis = seen[v]
            if c.overflowed() or M > c.m_cap or n_vis > c.nvis_cap:
                if raise_on_overflow:
                    raise L.BdsError(f"view {v}: list counts (M = {M}, visible = {n_vis}) exceed the calibrated capacities "
                                     f"({c.m_cap}, {c.nvis_cap}) right after calibration")
                ok = False
                self._grow(v, M, n_vis)
        if self.dynamic and not ok:
            self._grow_all_slots(max(m for m, _ in seen), max(n for _, n in seen))
        if self.fx is not None:      # the ranks' unions (identical counts on every rank)
            for v, n in enumerate(self.fx.static_counts()):
                self._unions[v] = max(self._unions[v], n)
                if n > self.fx.cap:
                    if raise_on_overflow:
                        raise L.BdsError(f"view {v}: union of the ranks' visible sets ({n}) exceeds the exchange capacity ({self.fx.cap})")
                    ok = False
        return ok

    def _grow_all_slots(self, M: int, n_vis: int) -> None:
        """dynamic: any slot may render any camera, so what one slot needed every slot is sized for (one recapture, not V of them)."""
        for v in range(self.V):
            self._grow(v, M, n_vis)
        m_cap, nv_cap = max(c.m_cap for c in self.caps), max(c.nvis_cap for c in self.caps)
        self.caps = [ListCapacity(m_cap, nv_cap) for _ in range(self.V)]

    def _agree(self, overflowed: bool, wants_more: bool):
        """The ranks' decision: (any rank overflowed, any rank wants larger lists).  A list count is rank-local -- one rank capturing
        again while its peers step on would pair its warm-up frame's collectives with their frame's."""
        if not self.coll:
            return overflowed, wants_more
        import torch.distributed as dist
        from .dist import _all_reduce
        flags = torch.tensor([int(overflowed), int(wants_more)], device=self.dev, dtype=torch.int32)
        _all_reduce(flags, dist.ReduceOp.MAX)
        o, w = flags.tolist()
        return bool(o), bool(w)

    def valid(self) -> bool:
        """Wait for the frame in flight; True if every view's lists (and exchange buffers) fitted on every rank.  Otherwise the
        capacities are grown, the graphs captured again and False is returned: the frame's gradients are incomplete (the overflowing
        view rendered nothing) -- do not step the optimizer on them.  With an exchange a COLLECTIVE call (every rank, every frame)."""
        for vg in self.views:
            vg.done.synchronize()
        if self.dynamic and any(int(e[0]) for e in self.sel_err):
            bad = [v for v, e in enumerate(self.sel_err) if int(e[0])]
            for e in self.sel_err:
                e.zero_()
            raise L.BdsError(f"view slot(s) {bad}: image index outside the {self.grids[0].shape[0]} bilateral grids (set_view's device-side "
                             f"index): the slot kept its previous grids and added no grid gradient -- the frame is not usable")
        ok = self._check_counts()
        # keep ahead of a growing scene: re-provision when a count comes within 8 % of its capacity -- at the start of the NEXT step
        # (the caller consumes this frame's gradients first; a capture's warm-up frame would overwrite the static outputs)
        grow = [v for v, c in enumerate(self.caps) if c.observed()[0] > 0.92 * c.m_cap or (c.observed()[1] > 0.92 * c.nvis_cap and c.nvis_cap < self.N)]
        near = self.fx is not None and max(self._unions) > 0.92 * self.fx.cap and self.fx.cap < (self.N + 3) // 4 * 4
        overflowed, wants_more = self._agree(not ok, bool(grow) or near)
        if overflowed:
            self.capture()
            self._stale = not self.clear_grads
            return False
        if wants_more:
            seen = [c.observed() for c in self.caps]
            for v in grow:
                self._grow(v, *seen[v])
            if self.dynamic and grow:
                self._grow_all_slots(max(m for m, _ in seen), max(n for _, n in seen))
            self._reprovision = True
        return True
