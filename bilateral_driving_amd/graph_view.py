"""A training frame as hipGraphs: every view's forward + loss + backward replayed with ONE launch each way, no host wait inside.

The reference's step has one host read-back per view (the intersection count that sizes gsplat's lists,
/root/reference/project/models/trainers/base.py:393-408) and ~100 framework launches around it.  Here the view runs in its
DEVICE-COUNT form (include/bds.h: list capacities from the host, actual counts read on the device) inside captured hipGraphs:

    frame = FrameGraph(params, cams, grids, skies, targets, factors)     # calibrates capacities, captures the graphs
    for it in range(n_iters):
        if frame.step():        # the frame's gradients are complete (False: a list outgrew its capacity; the graphs were captured
            optimizer.step()    # again with larger lists -- skip this frame's gradients and go on)

``step()`` waits for the frame and returns its validity, so that an overflowing view can not go unnoticed into an optimizer step;
``step(wait=False)`` only enqueues (benchmarks that pipeline frames call ``valid()`` themselves before they trust the gradients).

Fixed views (default): the cameras, targets, skies and image indices given at construction are what every replay renders.
Replayable views (``dynamic=True``) are what the reference's loop needs -- it draws a random image every step
(tools/train.py:250-283, datasets/base/split_wrapper.py:29): each of the V view slots owns STATIC input tensors (view matrix,
intrinsics, camera centre, target, sky, image index) that ``set_view(v, cam, target, sky, img_idx)`` overwrites before a replay;
the bilateral grids of the slot's image are picked on the device (bds_bilagrid_select), and the list capacities are the maximum over
a calibration sweep of ``calib_cams`` (the training cameras of this resolution) x ``headroom``.  One FrameGraph per (W, H) of the
reference's resolution schedule (trainers/base.py:142-146).

Gradients land exactly where the eager frame loop puts them: every ``param.grad`` is a slice of ONE flat buffer
(``dist.FlatGradients``), a view writes only the rows of the Gaussians it sees, the grids' gradients (transform + TV) are added in
place, ``sky.grad`` / ``viewmat.grad`` are the graphs' static outputs (``viewmat.grad`` of all views: one reduction of the views'
pose-gradient slots per frame; read it after ``step()``).  The loss value stays a slotted accumulator (``ViewGraph.loss`` sums it on
demand -- before the next ``step()``, whose record pack clears it), the dense screen-space gradient arrays behind
``info["means2d"].grad / .absgrad`` are per-view persistent buffers (``g2d``) whose stale rows the begin stage clears by the
previous visit's list.  Same kernels, same order, same numbers as ``harness.train_view`` -- tested equal, and against the oracle.

With an ``exchange`` (``dist.FrameExchange``; one process per GPU, every rank its own frame) the per-view collectives -- MAX
all-reduce of the visibility mask after the forward, SUM all-reduce of the compact gradient rows after the Gaussian half -- are
enqueued BETWEEN the graphs.  ``valid()`` / ``recapture()`` are then COLLECTIVE calls: the ranks agree (one MAX all-reduce of their
flags) before any of them captures again, because a capture's warm-up frame issues collectives of its own.
``FrameExchange(per_view=False)``: nothing between the graphs -- the frame replays exactly as on one GPU (both streams, in-place
accumulation) and ``step()`` ends with ONE dense all-reduce of the flat gradient buffer on the caller's stream; the next frame
clears that buffer densely (other ranks' rows are in it).  ``dist.plan_exchange`` prices the two modes.

A graph holds device addresses: after anything that moves a parameter's VALUES far enough to change the lists' sizes call
``recapture()``; after anything that RE-ALLOCATES the parameters (densification: other tensors, another N) call ``rebind(params)``
-- the same frame (cameras, targets, options) over the new tensors, with a gradient buffer and lists of the new size.  Slots,
capacities and the overflow protocol (a view whose lists outgrow their capacities renders NOTHING; ``valid()`` sees it after the
fact, grows the capacities and captures again): ``graph_slots.FrameCapacities``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from . import _lib as L
from .dist import ROW_NAMES, FlatGradients
from .fused_view import LIST_TILE
from . import harness as Hn
from .graph_slots import BLOCK_BOUNDS, GRAD_ROWS, FrameCapacities, ViewGraph, camera_centre as _camera_centre


class FrameGraph(FrameCapacities):
    def __init__(self, params: Dict[str, Tensor], cams: Sequence[Hn.Camera], grids: Sequence[Tensor], skies: Sequence[Tensor],
                 targets: Sequence[Tensor], factors: Sequence[int] = Hn.FACTORS_3, tv_weight: float = 0.01,
                 img_indices: Optional[Sequence[int]] = None, headroom: float = 1.5, list_tile: Optional[int] = None,
                 sh_degree: int = 3, extra_params: Sequence[Tensor] = (), overlap: bool = True, exchange=None, dynamic: bool = False,
                 calib_cams: Optional[Sequence[Hn.Camera]] = None, clear_grads: bool = True, row_catchup=None):
        """params: the five per-Gaussian leaves (``dist.ROW_NAMES``); grids: per level [n_img,12,L,gy,gx] leaves; cams / skies /
        targets: one per view of the frame; ``img_indices[v]``: the grid image of view v (default v).  ``headroom``: list capacity =
        headroom x the counts of the calibration visit.
        ``overlap`` (V > 1): every view is captured as two graphs and ``step()`` replays the forwards on a second stream next to the
        previous view's backward (round 3: one stream 782 it/s, two 885; a forward reads only parameters and writes its own buffers).
        ``dynamic``: replayable views (module docstring); ``calib_cams``: the cameras the capacities are sized over (default: ``cams``).
        ``exchange``: a ``dist.FrameExchange`` over ``params`` + ``grids`` (multi-GPU).
        ``clear_grads=False``: the begin stage does not clear the parameters' gradient rows (221 us per six-view frame at 2 M
        Gaussians) nor the dense tail -- the caller's optimizer does, as it consumes them (``optim.FusedAdam(consume_grads=True)``
        over EVERY parameter of the frame); a frame that ``step()`` reported invalid is cleared densely before the next one.
        ``row_catchup``: ``optim.DeferredRowAdam.catchup`` -- every view's forward brings the SH rows of its visible Gaussians up to
        the optimizer's step before its record pack reads them; step the optimizer with ``opt.step(lists=frame.row_lists())``."""
        assert sorted(params.keys()) == sorted(ROW_NAMES), "params: means, quats, log_scales, opacity_logits, sh"
        # (what rebind() needs to make this frame again over other parameter tensors)
        self._ctor = dict(cams=cams, grids=grids, skies=skies, targets=targets, factors=factors, tv_weight=tv_weight, img_indices=img_indices,
                          headroom=headroom, list_tile=list_tile, sh_degree=sh_degree, extra_params=extra_params, overlap=overlap,
                          exchange=exchange, dynamic=dynamic, calib_cams=calib_cams, clear_grads=clear_grads, row_catchup=row_catchup)
        self.params = {k: params[k] for k in ROW_NAMES}
        self.grids = list(grids)
        self.V = len(cams)
        assert len(skies) == self.V and len(targets) == self.V
        self.dev = self.params["means"].device
        self.dynamic = bool(dynamic)
        self.clear_grads = bool(clear_grads) or exchange is not None
        self.row_catchup = row_catchup
        self._stale = False              # clear_grads=False: the last frame was invalid, nobody consumed (and cleared) its gradients
        self.img_indices = list(range(self.V)) if img_indices is None else [int(i) for i in img_indices]
        if self.dynamic:       # every slot owns its inputs at fixed addresses; set_view() rewrites them
            self.cams = [Hn.Camera(c.viewmat.detach().clone().requires_grad_(c.viewmat.requires_grad), c.K.detach().clone(), c.width, c.height,
                                   (c.cam_pos if c.cam_pos is not None else _camera_centre(c.viewmat)).detach().clone()) for c in cams]
            self.skies = [s.detach().clone().requires_grad_(s.requires_grad) for s in skies]
            self.targets = [t.detach().clone() for t in targets]
            self.img_idx_dev = [torch.tensor([i], device=self.dev, dtype=torch.int32) for i in self.img_indices]
            # staging copies of ONE image's grids per slot (+ their gradients, scattered back by bds_bilagrid_select_bwd)
            self.sel_grids = [[torch.zeros(1, *g.shape[1:], device=self.dev, dtype=torch.float32).requires_grad_(True) for g in self.grids]
                              for _ in range(self.V)]
            self.sel_grads = [[torch.zeros(1, *g.shape[1:], device=self.dev, dtype=torch.float32) for g in self.grids] for _ in range(self.V)]
            self.sel_err = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(self.V)]   # sticky: image index out of range
        else:
            self.cams, self.skies, self.targets = list(cams), list(skies), list(targets)
        self.calib_cams = list(calib_cams) if calib_cams is not None else list(cams)
        assert all(c.width == self.cams[0].width and c.height == self.cams[0].height for c in list(self.cams) + self.calib_cams), \
            "one FrameGraph per image size"
        self.overlap = bool(overlap) and self.V > 1     # (one view: nothing to run a forward next to; c2: 687 vs 664 it/s with two graphs)
        self.factors, self.tv_weight, self.sh_degree = tuple(int(f) for f in factors), float(tv_weight), int(sh_degree)
        self.headroom = float(headroom)
        self.list_tile = int(LIST_TILE if list_tile is None else list_tile)
        L.require_gpu(*self.params.values(), *self.grids)
        self.N, self.K = self.params["means"].shape[0], self.params["sh"].shape[1]
        self.names = list(ROW_NAMES) + [f"grid{i}" for i in range(len(self.grids))]
        self.fx = exchange if (exchange is not None and exchange.active) else None
        self.world = exchange.world if exchange is not None else 1
        self.coll = bool(exchange is not None and exchange.coll)     # collectives are issued (several ranks, or one with dist.force_collectives)
        # FrameExchange(per_view=False) at world size > 1: the frame runs exactly as on one GPU and step() ends with ONE dense all-reduce
        self._frame_fx = exchange if (exchange is not None and exchange.frame_reduce) else None
        if exchange is not None:
            by_name = dict(self.params, **{f"grid{i}": g for i, g in enumerate(self.grids)})
            assert not extra_params and sorted(exchange.names) == sorted(self.names), "the exchange covers exactly params + grids"
            assert all(a is by_name[n] for a, n in zip(exchange.flat.params, exchange.names)), "the exchange was built over other tensors"
            self.names = list(exchange.names)
            self.flat, self.arena = exchange.flat, exchange.arena
        else:
            self.flat = FlatGradients(list(self.params.values()) + self.grids + list(extra_params), sparse_rows=True, row_block=GRAD_ROWS)
            self.arena = self.flat.arena(self.names + [f"extra{i}" for i in range(len(extra_params))])
        n_row = sum(self.arena[k].numel() for k in ROW_NAMES) + (5 * self.N if getattr(self.flat, "row_block", False) else 0)
        self._tail = self.flat.flat[n_row:]
        self.caps, self.views = [None] * self.V, [None] * self.V      # fused_view.ListCapacity / ViewGraph per view slot
        self.begin_graph = self.begin_graph_rest = None
        self.n_captures = 0
        self._reprovision = False        # grow + capture again at the start of the next step()
        lib = L.lib()
        self._ws_bytes = int(lib.bds_isect_prepare_workspace_bytes(1, self.N))
        self._ids_off = int(lib.bds_isect_visible_ids_offset(1, self.N))
        self._nvis_off = int(lib.bds_isect_counts_offset(3))
        # caller-owned prepare workspaces: a view's visible-id list and its counts live here from one frame to the next (the next
        # frame's begin graph clears exactly those gradient rows).  Zero-initialised: "no rows yet".
        self.prep_ws = [torch.zeros(max(self._ws_bytes, 16), device=self.dev, dtype=torch.uint8) for _ in range(self.V)]
        # per view: d(loss)/d(means2d) and its absolute sum, [2, N, 2], dense for the densification statistics (info["means2d"].grad /
        # .absgrad); the projection backward stores the visible rows, the begin stage clears the rows of the previous visit
        self.g2d = [torch.zeros(2, self.N, 2, device=self.dev, dtype=torch.float32) for _ in range(self.V)] if self.fx is None else None
        # all views' camera-pose gradient slots (+ loss accumulators) in ONE tensor: the slots are summed once per frame, not per view
        self._tail_rows = L.POSE_GRAD_SLOTS + L.LOSS_SLOTS * L.LOSS_SLOT_STRIDE // L.GRAD_RECORD_FLOATS
        self._tails = torch.zeros(self.V, self._tail_rows, L.GRAD_RECORD_FLOATS, device=self.dev, dtype=torch.float32)
        self._vm = torch.zeros(self.V, 4, 4, device=self.dev, dtype=torch.float32)
        self._defer_pose = self.fx is None and all(c.viewmat.requires_grad for c in self.cams)
        self._unions = [0] * self.V
        self.split_len, self.split_cap = [0] * self.V, [0] * self.V   # per slot: long tiles composited by four waves (graph_slots.calibrate)
        # a bound per 256-row block, refreshed by the begin stage: a view's projection skips the blocks it can not see (densify.spatial_order)
        self._bounds = torch.zeros((self.N + 255) // 256, 8, device=self.dev) if BLOCK_BOUNDS and self.N > 0 else None
        self.calibrate()
        self.capture()

    # ---- the phases of a view (eager warm-up, capture and replay walk the same protocol) -------------------------------------------
    def _view_kwargs(self, v: int) -> dict:
        kw = dict(factors=self.factors, tv_weight=self.tv_weight, caps=self.caps[v], prep_ws=self.prep_ws[v], list_tile=self.list_tile,
                  sh_degree=self.sh_degree, two_phase=True, lazy_loss=True, split_len=self.split_len[v], split_cap=self.split_cap[v],
                  # two streams: the transform's memory-bound last stage hides behind the other stream's compositor; folded into the
                  # compositor's backward it would lengthen the VALU-bound critical kernel (fused_view._DEFER_EPILOGUE)
                  defer_epilogue=not self.overlap, block_bounds=self._bounds, row_catchup=self.row_catchup)
        if self.fx is not None:     # rows into view v's compact exchange buffer; the dense tail (grids) accumulates in place in .grad
            kw.update(grad_sink=self.fx.static_sink(v), grid_grads=None)
            if self.dynamic:        # the transform's grid gradient -> the slot's staging slices, the TV term -> the parameters' slices
                kw.update(grid_grads=self.sel_grads[v], grad_arena={f"grid{i}": g for i, g in enumerate(self.sel_grads[v])},
                          tv_grids=self.grids, tv_grid_grads=[self.arena[f"grid{i}"] for i in range(len(self.grids))])
        else:
            grid_grads = [self.arena[f"grid{i}"] for i in range(len(self.grids))]
            arena = self.arena
            if self.dynamic:        # the transform's grid gradient goes to the slot's staging slices, the TV term to the parameters'
                arena = dict(self.arena, **{f"grid{i}": g for i, g in enumerate(self.sel_grads[v])})
                kw.update(tv_grids=self.grids, tv_grid_grads=grid_grads)
                grid_grads = self.sel_grads[v]
            kw.update(grid_grads=grid_grads, grad_arena=arena, arena_rows=1 if v == 0 else 2, g2d_buf=self.g2d[v])
            if self._defer_pose:
                kw.update(tail_buf=self._tails[v], defer_pose_sum=True)
        return kw

    def _phase_fwd(self, v: int):
        if self.dynamic:
            self._grid_select(v, bwd=False)
            for g, a in zip(self.sel_grids[v], self.sel_grads[v]):
                g.grad = a
            out = Hn.train_view(self.params, self.cams[v], self.sel_grids[v], None, self.skies[v], self.targets[v], **self._view_kwargs(v))
        else:
            out = Hn.train_view(self.params, self.cams[v], self.grids, self.img_indices[v], self.skies[v], self.targets[v],
                                **self._view_kwargs(v))
        if self.fx is not None:
            out["union_mask"] = (out["radii"].reshape(-1) > 0).to(torch.uint8)
        return out

    def _phase_bwd(self, v: int, out, tail: bool) -> None:
        out["backward"]()
        if self.dynamic:
            self._grid_select(v, bwd=True)
        if tail:
            out["backward_tail"]()

    def _point_grads_at_flat(self) -> None:
        for p, view in zip(self.flat.params, self.flat._views):
            p.grad = view
        if self.fx is None:
            # the flat buffer's own row book-keeping is bypassed (the begin graph clears the rows): make a later flat.zero() dense
            self.flat._dirty, self.flat._clean = None, False

    def _begin_body(self, views=None, tail: bool = True) -> None:
        """Row-wise clear of the gradient rows the previous frame's ``views`` (default: all) wrote + (``tail``) the dense tail."""
        lib, st, a = L.lib(), L.stream(), self.arena
        if self._bounds is not None and (views is None or 0 in views):      # (the parameters moved since the last frame)
            L.check(lib.bds_gaussian_block_bounds(self.N, L.ptr(self.params["means"]), L.ptr(self.params["log_scales"]), L.ptr(self._bounds), st),
                    "bds_gaussian_block_bounds")
        if self.fx is not None:      # the exchange's book of reduced rows clears the dense rows (fx.begin_frame); here only the tail
            if self._tail.numel() and tail:
                self._tail.zero_()
            return
        for v in (range(self.V) if views is None else views):
            ws = self.prep_ws[v]
            ids = ws[self._ids_off:self._ids_off + 4 * self.caps[v].nvis_cap].view(torch.int32)
            g2 = self.g2d[v]      # (the view's persistent screen-space gradient arrays: the same rows, no dense fill per view)
            row_clear = self.clear_grads and self._frame_fx is None   # (per-frame exchange: step() clears the flat buffer densely)
            pa = [a[k].data_ptr() if row_clear else None for k in ("means", "quats", "log_scales", "opacity_logits", "sh")]
            L.check(lib.bds_view_grads_clear_list_dev(self.caps[v].nvis_cap, ws.data_ptr() + self._nvis_off, L.ptr(ids), self.K, *pa,
                                                      L.ptr(g2[0]), L.ptr(g2[1]), st), "bds_view_grads_clear_list_dev")
        if self._tail.numel() and tail and self.clear_grads and self._frame_fx is None:
            self._tail.zero_()

    def _frame_begin(self) -> None:
        """Host side of a frame's start with an exchange: its row-wise clear of the dense gradient rows (eager: static id lists)."""
        if self.fx is not None:
            self.fx.begin_frame()
            self._point_grads_at_flat()

    # ---- capture -----------------------------------------------------------------------------------------------------------------
    def _capturing(self, graph, pool):
        """``torch.cuda.graph`` for this frame's captures.  With an exchange other threads of the process issue HIP calls of their own
        while we capture (RCCL's proxy, the process group's watchdog): only THIS thread's unsafe calls may invalidate the capture."""
        from .dist import _active
        if self.coll or _active():
            return torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local")
        return torch.cuda.graph(graph, pool=pool)

    def _reset_static_grads(self) -> None:
        for v in range(self.V):
            self.skies[v].grad = None
            self.cams[v].viewmat.grad = self._vm[v] if self._defer_pose else None   # (deferred: written by _sum_pose_slots)

    def capture(self) -> None:
        """(Re-)capture the begin graph and the view graphs against the current parameter / camera tensors and capacities.  With an
        exchange a COLLECTIVE call: its eager warm-up frame all-reduces (every rank must capture when one does)."""
        self.views = [None] * self.V
        self.begin_graph = self.begin_graph_rest = None
        self._reprovision = False
        torch.cuda.synchronize()
        if self.fx is not None:
            self.fx.static_setup(self._unions)
        self._point_grads_at_flat()
        self._reset_static_grads()
        # One eager frame in the device-count form on a side stream: lazy one-time work (kernel attributes, allocator growth) happens
        # here and not inside a capture, and it leaves real lists + counts in the prepare workspaces
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self._frame_begin()
            self._begin_body()
            for v in range(self.V):
                out = self._phase_fwd(v)
                if self.fx is not None:
                    self.fx.static_begin_view(v, out["union_mask"])
                self._phase_bwd(v, out, tail=False)
                if self.fx is not None:
                    self.fx.static_targets(v)
                out["backward_tail"]()
                if self.fx is not None:
                    self.fx.static_end_view(v)
            if self.fx is not None:
                self.fx.static_end_frame()
            self._reset_static_grads()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize()
        if not self.clear_grads:      # (the warm-up frame's gradients: nobody will consume them)
            self.flat.flat.zero_()
        self._check_counts(raise_on_overflow=True)
        self._point_grads_at_flat()
        # graphs that share a pool are replayed in capture order on ONE stream; the forwards of the overlapped form run on their own
        # stream and therefore get their own pool (a block one graph frees may be handed to the next graph of the same pool)
        self.pool = torch.cuda.graph_pool_handle()
        self.pool_fwd = torch.cuda.graph_pool_handle() if self.overlap else self.pool
        # HIGH priority for the forwards' stream: its kernels are mostly small and latency-bound (the tile stage's ~20 launches); at
        # equal priority their workgroups queue behind the thousands of pending workgroups of the other stream's compositor backward
        # (measured with in-graph timing marks, scripts/overlap_timeline.py: the 60 us list build took 370 us next to it)
        self.side_stream = torch.cuda.Stream(device=self.dev, priority=-1) if self.overlap else None
        self._frame_ready, self._frame_ready_rest = torch.cuda.Event(), torch.cuda.Event()
        outer, L.GRAPH_MARKS = L.GRAPH_MARKS, {}     # timing marks captured into THESE graphs (when _lib timers are enabled)
        try:
            g = torch.cuda.CUDAGraph()
            if self.overlap and self.fx is None:
                # two begin graphs: the forwards' stream only has to wait until the begin stage has READ view 0's old id list (which
                # view 0's forward overwrites); the other clears (~100 us at 2 M Gaussians) run next to that forward
                with self._capturing(g, self.pool):
                    self._begin_body(views=[0], tail=True)
                g2 = torch.cuda.CUDAGraph()
                with self._capturing(g2, self.pool):
                    self._begin_body(views=range(1, self.V), tail=False)
                self.begin_graph_rest = g2
            else:
                with self._capturing(g, self.pool):
                    self._begin_body()
            self.begin_graph = g
            # ALL forwards first, then all backwards: a block of the forwards' pool that a backward's capture frees (buffers the
            # forward prepared for it) could otherwise be handed to the NEXT view's forward, which runs next to that backward
            fwd, bwd = [], []
            for v in range(self.V):
                g = torch.cuda.CUDAGraph()
                with self._capturing(g, self.pool_fwd):
                    out = self._phase_fwd(v)
                fwd.append((g, out))
            split = self.fx is not None      # the Gaussian half is a graph of its own where the exchange goes between the halves
            for v, (g, out) in enumerate(fwd):
                gb = torch.cuda.CUDAGraph()
                with self._capturing(gb, self.pool):
                    self._phase_bwd(v, out, tail=not split)
                bwd.append(gb)
            for v, (g, out) in enumerate(fwd):
                gt = None
                if split:
                    gt = torch.cuda.CUDAGraph()
                    with self._capturing(gt, self.pool):
                        out["backward_tail"]()
                self.views[v] = ViewGraph(g, out, self.caps[v], self.prep_ws[v], self.skies[v], self.cams[v].viewmat, bwd[v], gt)
        finally:
            self.marks, L.GRAPH_MARKS = L.GRAPH_MARKS, outer
        self.n_captures += 1
        torch.cuda.synchronize()

    def recapture(self) -> None:
        """The scene changed under the same tensors: size the lists again, then capture."""
        self.calibrate(); self.capture()

    def rebind(self, params: Dict[str, Tensor], **changes) -> None:
        """After densification (``densify.refinement_after``: the parameters are OTHER tensors of another length): this frame again --
        same cameras / targets / skies / options, ``changes`` overriding constructor arguments (e.g. ``row_catchup`` of a new
        optimizer) -- over ``params``: new flat gradient buffer, new per-view buffers, lists calibrated for the new set, new graphs.
        Single-process frames (an exchange holds a flat buffer of its own: build a new ``FrameExchange`` and pass ``exchange=``)."""
        kw = dict(self._ctor, **changes)
        assert kw.get("exchange") is None or "exchange" in changes, "rebind: pass the new exchange (its flat buffer is sized for the old parameters)"
        if self.dynamic:      # the slots' CURRENT inputs, not the ones of the first construction
            kw.update(cams=self.cams, skies=self.skies, targets=self.targets, img_indices=self.img_indices)
        torch.cuda.synchronize()
        self.__init__(params, **kw)

    # ---- replay ------------------------------------------------------------------------------------------------------------------
    def step(self, serial: bool = False, wait: bool = True, local: bool = False) -> Optional[bool]:
        """One frame: clear the previous frame's gradient rows, then every view (forward + loss + backward), gradients summed (over
        the ranks too, with an exchange).  ``wait`` (default): waits for the frame and returns ``valid()`` -- False: a list overflowed,
        the gradients are incomplete and must not reach the optimizer; with ``wait=False`` nothing is waited for and None is returned
        (call ``valid()`` before trusting the gradients).  ``serial``: every graph on the caller's stream, one after the other
        (measurement).  ``local``: leave out the per-frame all-reduce of ``FrameExchange(per_view=False)`` (timing one rank's compute)."""
        if self._reprovision:       # capacities that came close to their limit in the last (valid, consumed) frame
            self.capture()
        if self._stale:             # clear_grads=False and nobody consumed the invalid frame's gradients
            self.flat.flat.zero_()
            self._stale = False
        fx = self.fx
        main = torch.cuda.current_stream(self.dev)
        if self._frame_fx is not None:   # the last frame's all-reduce left the other ranks' rows behind: the row-wise clear does not know them
            self.flat.flat.zero_()
        self._frame_begin()
        self.begin_graph.replay()
        rest = self.begin_graph_rest
        side = None if (serial or not self.overlap) else self.side_stream
        if rest is not None and side is None:
            rest.replay()
            rest = None
        if side is not None:
            # the forwards start once everything enqueued so far (the previous frame's backwards, an optimizer step, set_view copies)
            # is done and the begin graph has read the previous frame's visible-id lists, which the forwards overwrite -- and then
            # run ahead of the backwards on their own stream
            self._frame_ready.record(main)
            side.wait_event(self._frame_ready)
            if rest is not None:
                rest.replay()
                self._frame_ready_rest.record(main)
            with torch.cuda.stream(side):
                for v, vg in enumerate(self.views):
                    if v == 1 and rest is not None:
                        side.wait_event(self._frame_ready_rest)     # (the other views' old id lists have been read)
                    vg.graph.replay()
                    if fx is not None:
                        fx.static_begin_view(v, vg.out["union_mask"])   # (RCCL orders the mask's all-reduce behind this stream)
                    vg.fwd_done.record(side)
        for v, vg in enumerate(self.views):
            if side is None:
                vg.graph.replay()
                if fx is not None:
                    fx.static_begin_view(v, vg.out["union_mask"])
            else:
                main.wait_event(vg.fwd_done)
            vg.graph_bwd.replay()       # with an exchange: the image half, which needs nothing from the other ranks
            if fx is not None:
                fx.static_targets(v)    # waits for the union mask; slot map + cleared buffer rows (two launches)
                vg.graph_tail.replay()  # the Gaussian half: the only writer of the per-Gaussian gradient rows
                fx.static_end_view(v)   # all-reduce of view v's rows (async); reduced rows of view v - 1 added to the dense buffer
                if v == self.V - 1:
                    fx.static_end_frame()
            vg.done.record(main)
        self._sum_pose_slots()
        if self._frame_fx is not None and not local:
            self._frame_fx.reduce_frame()     # (on this stream: whoever consumes the gradients on it is ordered behind the collective)
        return self.valid() if wait else None

    def _sum_pose_slots(self) -> None:
        """Camera-pose gradients of all views: ONE reduction of the views' slots per frame (``viewmat.grad`` = a row of the result)."""
        if self._defer_pose:
            torch.sum(self._tails[:, :L.POSE_GRAD_SLOTS].view(self.V, L.POSE_GRAD_SLOTS, 4, 4), dim=1, out=self._vm)

    def mark_samples(self, name: str):
        """Milliseconds of every timing mark pair ``name`` captured into the view graphs (``_lib.enable_timers`` on during the
        capture), as recorded by the last replay; synchronise first."""
        return L.graph_mark_samples(name, getattr(self, "marks", {}))
