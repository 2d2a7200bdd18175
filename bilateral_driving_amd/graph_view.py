"""A training frame as hipGraphs: every view's forward + loss + backward replayed with ONE launch, no host wait anywhere.

The reference's step has one host read-back per view (the intersection count that sizes gsplat's lists,
/root/reference/project/models/trainers/base.py:393-408) and ~100 framework launches around it.  ``fused_view`` already folds the
launches into ~40 libbds kernels behind one host wait; on a slow or busy host that wait and the Python between the launches still
leave the GPU idle for a quarter of the step.  Here the view runs in its DEVICE-COUNT form (include/bds.h: list capacities from the
host, actual counts read on the device) inside a captured hipGraph:

    frame = FrameGraph(params, cams, grids, skies, targets, factors)     # calibrates capacities, captures V + 1 graphs
    for it in range(n_iters):
        frame.step()            # begin graph (row-wise clear of the previous frame's gradient rows) + one graph per view
        if not frame.valid():   # optional before an optimizer step: waits for the frame, re-captures with larger lists and
            continue            # returns False if a list outgrew its capacity (the frame's gradients are then to be discarded)
        optimizer.step()

Gradients land exactly where the eager frame loop (``bench.py`` / ``dist.FrameExchange`` at world size 1) puts them: every
``param.grad`` is a slice of ONE flat buffer (``dist.FlatGradients``), a view writes only the rows of the Gaussians it sees, the grids'
gradients (transform + TV) are added in place, ``sky.grad`` / ``viewmat.grad`` are the graph's static outputs.  Same kernels, same
order, same numbers as ``harness.train_view`` -- tested equal.

What the replayed frame keeps OUT of its launch list (a launch costs ~4.5 us of GPU time even when it does nothing, and the frame's time
is the sum of its kernels' times whatever runs next to what -- DESIGN.md "The step"): the loss value is left as a slotted accumulator
(``ViewGraph.loss`` sums it when asked), ``viewmat.grad`` of every camera is a row of one tensor filled by ONE reduction of all views'
pose-gradient slots per frame (``_sum_pose_slots``; read it after ``step()`` on the stepping stream), the dense screen-space gradient
arrays behind ``info["means2d"].grad / .absgrad`` are per-view persistent buffers (``g2d``) whose stale rows the begin stage clears by
the previous visit's list.

A graph holds device addresses: after anything that re-allocates a parameter (densification) or changes a camera call
``frame.recapture()``.  Overflow protocol: a view whose list counts outgrow their capacities renders NOTHING (effective counts zero,
see bds_isect_prepare_dev) and raises the sticky overflow word in its page-locked counts; ``valid()`` / ``check()`` see it after the
fact, grow the capacities (never shrink) and capture again.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from . import _lib as L
from .dist import ROW_NAMES, FlatGradients
from .fused_view import LIST_TILE, ListCapacity
from . import harness as Hn


# A/B switch (measurement): 0 = the loss accumulator filled and summed, and the screen-space gradient arrays filled, by launches of
# their own in every view (three more launches per view)
_FEWER_LAUNCHES = os.environ.get("BDS_FEWER_LAUNCHES", "1") == "1"


class ViewGraph:
    """One captured view: forward, L1 + TV loss and backward of camera ``cam`` with image index ``img_idx`` -- ONE graph, or two
    (``graph`` = forward + loss value, ``graph_bwd`` = the rest) when the frame overlaps a view's forward with the previous view's
    backward."""

    def __init__(self, graph, out, caps: ListCapacity, prep_ws: Tensor, sky: Tensor, viewmat: Tensor, graph_bwd=None, graph_tail=None,
                 graph_front=None):
        self.graph, self.graph_bwd, self.graph_tail, self.out, self.caps, self.prep_ws = graph, graph_bwd, graph_tail, out, caps, prep_ws
        self.graph_front = graph_front   # (front_stream: projection + lists captured apart from the compositor / colour transform)
        self.front_done = torch.cuda.Event()
        # static tensors: rewritten by every replay.  The loss VALUE is left as its slotted accumulator when the loss rode on the colour
        # transform's launch (``loss`` sums the 64 slots on demand: one launch less per view in the replayed frame)
        self._loss, self._loss_slots = out.get("loss"), out.get("loss_slots")
        self.rgb, self.depth, self.opacity = out["rgb"], out["depth"], out["opacity"]
        self.v_sky, self.v_viewmat = sky.grad, viewmat.grad
        self.done = torch.cuda.Event()
        self.fwd_done, self.bwd_done = torch.cuda.Event(), torch.cuda.Event()

    @property
    def loss(self) -> Tensor:
        if self._loss_slots is None:
            return self._loss
        from .losses import slots_value
        torch.cuda.current_stream(self._loss_slots.device).wait_event(self.done)
        return slots_value(self._loss_slots)

    def replay(self) -> None:
        """All three graphs of the view on the current stream (world size 1: with an exchange the collectives go between them,
        ``FrameGraph.step``)."""
        if self.graph_front is not None:
            self.graph_front.replay()
        self.graph.replay()
        self.graph_bwd.replay()
        if self.graph_tail is not None:
            self.graph_tail.replay()
        self.done.record()


class FrameGraph:
    def __init__(self, params: Dict[str, Tensor], cams: Sequence[Hn.Camera], grids: Sequence[Tensor], skies: Sequence[Tensor],
                 targets: Sequence[Tensor], factors: Sequence[int] = Hn.FACTORS_3, tv_weight: float = 0.01,
                 img_indices: Optional[Sequence[int]] = None, headroom: float = 1.5, list_tile: Optional[int] = None,
                 sh_degree: int = 3, extra_params: Sequence[Tensor] = (), overlap: bool = True, overlap_tail: bool = False,
                 exchange=None, bwd_streams: int = 1, fork_tail: bool = False, late_image: bool = False, front_stream: bool = False, single_graph: bool = False,
                 phase_shift: bool = False):
        """params: the five per-Gaussian leaves (``dist.ROW_NAMES``); grids: per level [n_img,12,L,gy,gx] leaves; cams / skies /
        targets: one per view of the frame; ``img_indices[v]``: the grid image of view v (default v).  ``headroom``: list capacity =
        headroom x the counts of the calibration visit.
        ``overlap``: every view is captured as THREE graphs (forward + loss value | image half of the backward | Gaussian half) and
        ``step()`` replays them on three streams: view v + 1's forward -- projection, the launch-latency-bound tile stage, SH, the
        compositor's forward, the gather-bound bilateral forward -- and view v - 1's list-driven SH / projection backward run next to
        view v's bilateral + compositor backward (VALU-bound) instead of in front of / behind it.  A forward reads only parameters and
        writes its own buffers; the image halves follow one another on the caller's stream, the Gaussian halves -- the only writers of
        the per-Gaussian gradient rows -- on the third when ``overlap_tail`` is set, else behind their image halves on the caller's
        stream (same accumulation order either way: same numbers).  Measured on MI355X, 2 M Gaussians / six 1080p views: one stream
        782 it/s, forwards on a second stream 885, Gaussian halves on a third 875 -- the default is two.
        ``bwd_streams`` > 1 (implies ``overlap_tail``): the image halves of consecutive views alternate between that many streams, so
        that view v + 1's bilateral backward (gather-latency-bound) runs next to view v's compositor backward (VALU-bound); the
        grids' gradient slices are accumulated with atomics for that.
        ``fork_tail``: inside the captured backward the SH half of the Gaussian backward forks onto another stream next to the
        projection half (a graph with parallel branches).  Measured: 772 vs 909 it/s without -- the HIP graph executor serialises badly
        around a fork / join; off.
        ``late_image``: True: the forward graph stops behind the compositor; "front": behind the tile lists (the compositor's forward
        moves too); the colour transform (expected depth, clamp, sky blend,
        bilateral transform) and the loss are captured in front of the backward's image half instead (on the backward's stream).
        ``exchange``: a ``dist.FrameExchange`` over ``params`` + ``grids`` (multi-GPU: one process per GPU, every rank its own frame).
        The per-view collectives -- MAX-all-reduce of the visibility mask after the forward, SUM-all-reduce of the compact gradient
        rows after the Gaussian half -- are enqueued BETWEEN the graphs (RCCL runs them on its own stream next to the following
        graphs); the slot map of the ranks' union is two libbds launches in front of the Gaussian half."""
        assert sorted(params.keys()) == sorted(ROW_NAMES), "params: means, quats, log_scales, opacity_logits, sh"
        self.params = {k: params[k] for k in ROW_NAMES}
        self.cams, self.grids, self.skies, self.targets = list(cams), list(grids), list(skies), list(targets)
        self.V = len(self.cams)
        assert len(self.skies) == self.V and len(self.targets) == self.V
        if self.V == 1:
            overlap = False      # nothing to run a forward next to: one graph per view (c2: 687 vs 664 it/s with two graphs on two streams)
        self.factors, self.tv_weight, self.sh_degree = tuple(int(f) for f in factors), float(tv_weight), int(sh_degree)
        self.img_indices = list(range(self.V)) if img_indices is None else [int(i) for i in img_indices]
        self.headroom = float(headroom)
        # front_stream: every view's projection + lists (no dependence on anything but the parameters) as a graph of its own on a THIRD
        # stream, running ahead of the compositors' stream
        # single_graph: the whole frame -- begin body, every forward on a forked branch, every backward behind its forward -- captured
        # as ONE hipGraph with two parallel branches: one launch per frame instead of 1 + 2 per view, no inter-graph gaps on either branch
        self.single_graph = bool(single_graph and overlap and exchange is None and not front_stream and not overlap_tail and bwd_streams <= 1)
        # phase_shift (implies front_stream and a Gaussian half captured apart): the compositor's forward of view v + 2 is held back until
        # the compositor's backward of view v has finished, so that each compositor runs next to the OTHER stream's gather-bound
        # kernels (bilateral, tile stage) instead of next to the other compositor
        self.phase_shift = bool(phase_shift and overlap and exchange is None)
        if self.phase_shift:
            front_stream, overlap_tail = True, True
        self.front_stream = bool(front_stream and overlap and exchange is None)
        self.late_image = "front" if self.front_stream else (late_image if late_image == "front" else bool(late_image))
        self.n_bwd_streams = max(1, int(bwd_streams)) if overlap else 1
        self.overlap, self.overlap_tail = bool(overlap), bool(overlap and (overlap_tail or self.n_bwd_streams > 1 or (phase_shift and exchange is None)))
        self.list_tile = int(LIST_TILE if list_tile is None else list_tile)
        self.dev = self.params["means"].device
        L.require_gpu(*self.params.values(), *self.grids)
        self.N, self.K = self.params["means"].shape[0], self.params["sh"].shape[1]
        self.names = list(ROW_NAMES) + [f"grid{i}" for i in range(len(self.grids))]
        self.fx = exchange if (exchange is not None and exchange.active) else None
        if exchange is not None:
            by_name = dict(self.params, **{f"grid{i}": g for i, g in enumerate(self.grids)})
            assert not extra_params and sorted(exchange.names) == sorted(self.names), "the exchange covers exactly params + grids"
            assert all(a is by_name[n] for a, n in zip(exchange.flat.params, exchange.names)), "the exchange was built over other tensors"
            self.names = list(exchange.names)
            self.flat, self.arena = exchange.flat, exchange.arena
        else:
            self.flat = FlatGradients(list(self.params.values()) + self.grids + list(extra_params), sparse_rows=True)
            self.arena = self.flat.arena(self.names + [f"extra{i}" for i in range(len(extra_params))])
        n_row = sum(self.arena[k].numel() for k in ROW_NAMES)
        self._tail = self.flat.flat[n_row:]
        self.caps: List[Optional[ListCapacity]] = [None] * self.V
        self.views: List[Optional[ViewGraph]] = [None] * self.V
        self.begin_graph = None
        self.pool = None
        self.n_captures = 0
        lib = L.lib()
        self._ws_bytes = int(lib.bds_isect_prepare_workspace_bytes(1, self.N))
        self._ids_off = int(lib.bds_isect_visible_ids_offset(1, self.N))
        self._nvis_off = int(lib.bds_isect_counts_offset(3))
        # caller-owned prepare workspaces: a view's visible-id list and its counts live here from one frame to the next (the next
        # frame's begin graph clears exactly those gradient rows).  Zero-initialised: "no rows yet".
        self.prep_ws = [torch.zeros(max(self._ws_bytes, 16), device=self.dev, dtype=torch.uint8) for _ in range(self.V)]
        # per view: d(loss)/d(means2d) and its absolute sum, [2, N, 2], dense for the densification statistics (info["means2d"].grad /
        # .absgrad); the projection backward stores the visible rows, the begin stage clears the rows of the previous visit
        self.g2d = ([torch.zeros(2, self.N, 2, device=self.dev, dtype=torch.float32) for _ in range(self.V)]
                    if not (exchange is not None and exchange.active) else None)
        # all views' camera-pose gradient slots (+ loss accumulators) in ONE tensor: the slots are summed once per frame, not per view
        self._tail_rows = L.POSE_GRAD_SLOTS + L.LOSS_SLOTS * L.LOSS_SLOT_STRIDE // L.GRAD_RECORD_FLOATS
        self._tails = torch.zeros(self.V, self._tail_rows, L.GRAD_RECORD_FLOATS, device=self.dev, dtype=torch.float32)
        self._vm = torch.zeros(self.V, 4, 4, device=self.dev, dtype=torch.float32)
        self._defer_pose = (_FEWER_LAUNCHES and not (exchange is not None and exchange.active)
                            and all(c.viewmat.requires_grad for c in self.cams))
        self._unions = [0] * self.V
        # (inside the captured backward: the SH half of the Gaussian backward forks onto this stream, see fused_view.backward_steps)
        self._fork_stream = torch.cuda.Stream(device=self.dev) if fork_tail else None
        self.calibrate()
        self.capture()

    # ---- capacities --------------------------------------------------------------------------------------------------------------
    def calibrate(self) -> None:
        """One forward visit of every camera through the host-count path: the list capacities (and, with an exchange, the size of the
        ranks' union per view) are sized from what it needed."""
        import torch.distributed as dist
        with torch.no_grad():
            for v, cam in enumerate(self.cams):
                info = Hn.render_view(self.params, cam, self.grids, self.img_indices[v], self.skies[v], factors=self.factors,
                                      sh_degree=self.sh_degree, list_tile=self.list_tile)["info"]
                self._grow(v, int(info["n_isects"]), int(info["n_visible"]))
                if self.fx is not None:
                    mask = (info["radii"].reshape(-1) > 0).to(torch.uint8)
                    if self.fx.world > 1:
                        dist.all_reduce(mask, op=dist.ReduceOp.MAX)
                    self._unions[v] = max(self._unions[v], int(mask.sum()))

    def _grow(self, v: int, M: int, n_vis: int) -> None:
        old = self.caps[v]
        h = self.headroom if old is None else max(self.headroom, 1.25)   # (a re-provision is a re-capture: make it worth it)
        m_cap = max(int(M * h) + 4096, old.m_cap if old else 0)
        nv_cap = max(min(int(n_vis * h) + 1024, self.N), old.nvis_cap if old else 0)   # never shrinks
        self.caps[v] = ListCapacity(m_cap, max(nv_cap, 1))

    # ---- the phases of a view (eager warm-up, capture and replay walk the same protocol) -------------------------------------------
    def _view_kwargs(self, v: int) -> dict:
        kw = dict(factors=self.factors, tv_weight=self.tv_weight, caps=self.caps[v], prep_ws=self.prep_ws[v], list_tile=self.list_tile,
                  sh_degree=self.sh_degree, two_phase=True, tail_fork_stream=self._fork_stream, late_image=self.late_image,
                  lazy_loss=_FEWER_LAUNCHES)
        if self.fx is not None:     # rows into view v's compact exchange buffer; the dense tail (grids) accumulates in place in .grad
            kw.update(grad_sink=self.fx.static_sink(v), grid_grads=None)
        else:
            kw.update(grid_grads=[self.arena[f"grid{i}"] for i in range(len(self.grids))], grad_arena=self.arena,
                      arena_rows=1 if v == 0 else 2, g2d_buf=self.g2d[v] if _FEWER_LAUNCHES else None)
            if self._defer_pose:
                kw.update(tail_buf=self._tails[v], defer_pose_sum=True)
        return kw

    def _phase_fwd(self, v: int):
        out = Hn.train_view(self.params, self.cams[v], self.grids, self.img_indices[v], self.skies[v], self.targets[v],
                            **self._view_kwargs(v))
        if self.fx is not None:
            out["union_mask"] = (out["radii"].reshape(-1) > 0).to(torch.uint8)
        return out

    def _point_grads_at_flat(self) -> None:
        for p, view in zip(self.flat.params, self.flat._views):
            p.grad = view
        if self.fx is None:
            # the flat buffer's own row book-keeping is bypassed (the begin graph clears the rows): make a later flat.zero() dense
            self.flat._dirty, self.flat._clean = None, False

    def _begin_body(self, views=None, tail: bool = True) -> None:
        """Row-wise clear of the gradient rows the previous frame's ``views`` (default: all) wrote + (``tail``) the dense tail."""
        if self.fx is not None:      # the exchange's book of reduced rows clears the dense rows (fx.begin_frame); here only the tail
            if self._tail.numel() and tail:
                self._tail.zero_()
            return
        lib, st = L.lib(), L.stream()
        a = self.arena
        for v in (range(self.V) if views is None else views):
            ws = self.prep_ws[v]
            ids = ws[self._ids_off:self._ids_off + 4 * self.caps[v].nvis_cap].view(torch.int32)
            g2 = self.g2d[v]      # (the view's persistent screen-space gradient arrays: the same rows, no dense fill per view)
            L.check(lib.bds_view_grads_clear_list_dev(self.caps[v].nvis_cap, ws.data_ptr() + self._nvis_off, L.ptr(ids), self.K,
                                                      L.ptr(a["means"]), L.ptr(a["quats"]), L.ptr(a["log_scales"]),
                                                      L.ptr(a["opacity_logits"]), L.ptr(a["sh"]), L.ptr(g2[0]), L.ptr(g2[1]), st),
                    "bds_view_grads_clear_list_dev")
        if self._tail.numel() and tail:
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
        if self.fx is not None:
            return torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local")
        return torch.cuda.graph(graph, pool=pool)

    def capture(self) -> None:
        """(Re-)capture the begin graph and the view graphs against the current parameter / camera tensors and capacities."""
        self.views = [None] * self.V
        self.begin_graph = None
        self.pool = None
        torch.cuda.synchronize()
        if self.fx is not None:
            self.fx.static_setup(self._unions)
        self._point_grads_at_flat()
        for v in range(self.V):
            self.skies[v].grad = None
            self.cams[v].viewmat.grad = self._vm[v] if self._defer_pose else None   # (deferred: written by _sum_pose_slots)
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
                out["backward"]()
                if self.fx is not None:
                    self.fx.static_targets(v)
                out["backward_tail"]()
                if self.fx is not None:
                    self.fx.static_end_view(v)
                self.skies[v].grad = None
                self.cams[v].viewmat.grad = self._vm[v] if self._defer_pose else None
            if self.fx is not None:
                self.fx.static_end_frame()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize()
        self._check_counts(raise_on_overflow=True)
        self._point_grads_at_flat()
        # graphs that share a pool are replayed in capture order on ONE stream; the forwards of the overlapped form run on their own
        # stream and therefore get their own pool (a block one graph frees may be handed to the next graph of the same pool)
        self.pool = torch.cuda.graph_pool_handle()
        self.pools_bwd = [self.pool] + [torch.cuda.graph_pool_handle() for _ in range(self.n_bwd_streams - 1)]
        self.extra_bwd_streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.n_bwd_streams - 1)]
        self.pool_fwd = torch.cuda.graph_pool_handle() if self.overlap else self.pool
        self.pool_front = torch.cuda.graph_pool_handle() if self.front_stream else None
        self.front_stream_h = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("BDS_FRONT_STREAM_PRIORITY", "0"))) \
            if self.front_stream else None
        self.pool_tail = torch.cuda.graph_pool_handle() if self.overlap_tail else self.pool
        # HIGH priority for the forwards' stream: its kernels are mostly small and latency-bound (the tile stage's ~20 launches); at
        # equal priority their workgroups queue behind the thousands of pending workgroups of the other stream's compositor backward
        # (measured with in-graph timing marks, scripts/overlap_timeline.py: the 60 us list build took 370 us next to it)
        prio = int(os.environ.get("BDS_FWD_STREAM_PRIORITY", "-1"))
        self.side_stream = torch.cuda.Stream(device=self.dev, priority=prio) if self.overlap else None
        self.tail_stream = torch.cuda.Stream(device=self.dev) if self.overlap_tail else None
        self._frame_ready, self._frame_ready_rest = torch.cuda.Event(), torch.cuda.Event()
        outer, L.GRAPH_MARKS = L.GRAPH_MARKS, {}     # timing marks captured into THESE graphs (when _lib timers are enabled)
        self.frame_graph = None
        try:
            if self.single_graph:
                self._capture_single()
                return
            g = torch.cuda.CUDAGraph()
            self.begin_graph_rest = None
            if self.overlap and self.fx is None and self.V > 1:
                # two begin graphs: the forwards' stream only has to wait until the begin stage has READ view 0's old id list (which
                # view 0's forward overwrites); the other five clears (~100 us at 2 M Gaussians) run next to that forward
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
            # ALL forwards first, then all image halves, then all Gaussian halves: a block of the forwards' pool that a later stage's
            # capture frees (buffers the forward prepared for it) could otherwise be handed to the NEXT view's forward, which runs
            # next to that stage (same rule one stage down)
            fwd, bwd, fronts = [], [], [None] * self.V
            if self.front_stream:
                outs = []
                for v in range(self.V):
                    gf = torch.cuda.CUDAGraph()
                    with self._capturing(gf, self.pool_front):
                        outs.append(self._phase_fwd(v))          # (late_image = "front": stops behind the lists)
                    fronts[v] = gf
                for v in range(self.V):
                    g = torch.cuda.CUDAGraph()
                    with self._capturing(g, self.pool_fwd):
                        outs[v]["image"]()                         # compositor, colour transform, loss value + gradient
                    fwd.append((g, outs[v]))
            else:
                for v in range(self.V):
                    g = torch.cuda.CUDAGraph()
                    with self._capturing(g, self.pool_fwd):
                        out = self._phase_fwd(v)
                    fwd.append((g, out))
            # (the Gaussian half is a graph of its own only where something goes between the halves: a third stream, or the exchange)
            split = self.overlap_tail or self.fx is not None
            for v, (g, out) in enumerate(fwd):
                gb = torch.cuda.CUDAGraph()
                with self._capturing(gb, self.pools_bwd[v % self.n_bwd_streams]):
                    out["backward"]()
                    if not split:
                        out["backward_tail"]()
                bwd.append(gb)
            for v, (g, out) in enumerate(fwd):
                gt = None
                if split:
                    gt = torch.cuda.CUDAGraph()
                    with self._capturing(gt, self.pool_tail):
                        out["backward_tail"]()
                self.views[v] = ViewGraph(g, out, self.caps[v], self.prep_ws[v], self.skies[v], self.cams[v].viewmat, bwd[v], gt, fronts[v])
        finally:
            self.marks, L.GRAPH_MARKS = L.GRAPH_MARKS, outer
        self.n_captures += 1
        torch.cuda.synchronize()

    def _capture_single(self) -> None:
        """The frame as one graph: the capture stream carries the begin body and the backwards, ``side_stream`` (forked from it inside
        the capture) the forwards.  Same ordering rule as the separate graphs: every forward is captured before the first backward, so
        no block a backward's capture frees can be handed to a forward that runs next to it."""
        G = torch.cuda.CUDAGraph()
        side = self.side_stream
        with self._capturing(G, self.pool):
            cs = torch.cuda.current_stream(self.dev)
            self._begin_body()
            fork = torch.cuda.Event()
            fork.record(cs)
            side.wait_event(fork)
            outs = []
            with torch.cuda.stream(side):
                for v in range(self.V):
                    out = self._phase_fwd(v)
                    e = torch.cuda.Event()
                    e.record(side)
                    outs.append((out, e))
            for v, (out, e) in enumerate(outs):
                cs.wait_event(e)
                out["backward"]()
                out["backward_tail"]()
            cs.wait_stream(side)
        self.frame_graph = G
        self.begin_graph = None
        for v, (out, _e) in enumerate(outs):
            self.views[v] = ViewGraph(None, out, self.caps[v], self.prep_ws[v], self.skies[v], self.cams[v].viewmat, None, None)
        self.n_captures += 1
        torch.cuda.synchronize()

    recapture = capture

    # ---- replay ------------------------------------------------------------------------------------------------------------------
    def step(self, serial: bool = False) -> None:
        """One frame: clear the previous frame's gradient rows, then every view (forward + loss + backward), gradients summed (over
        the ranks too, with an exchange).  ``serial``: every graph on the caller's stream, one after the other (measurement)."""
        fx = self.fx
        main = torch.cuda.current_stream(self.dev)
        self._frame_begin()
        if self.frame_graph is not None:     # single_graph: the two branches are inside (``serial`` has no meaning here)
            self.frame_graph.replay()
            self._sum_pose_slots()
            for vg in self.views:
                vg.done.record(main)
            return
        self.begin_graph.replay()
        rest = getattr(self, "begin_graph_rest", None)
        if self.phase_shift and not serial:
            self._step_phase_shifted(main, rest)
            return
        if rest is not None and (serial or not self.overlap):
            rest.replay()
            rest = None
        side = None if (serial or not self.overlap) else self.side_stream
        tail = None if (serial or not self.overlap_tail) else self.tail_stream
        nb = 1 if tail is None else self.n_bwd_streams
        if side is not None:
            # the forwards start once everything enqueued so far (the previous frame's backwards, an optimizer step) is done and the
            # begin graph has read the previous frame's visible-id lists, which the forwards overwrite -- and then run ahead of the
            # backwards on their own stream
            self._frame_ready.record(main)
            side.wait_event(self._frame_ready)
            if rest is not None:
                rest.replay()
                self._frame_ready_rest.record(main)
            if self.front_stream:
                fs = self.front_stream_h
                fs.wait_event(self._frame_ready)
                with torch.cuda.stream(fs):
                    for v, vg in enumerate(self.views):
                        if v == 1 and rest is not None:
                            fs.wait_event(self._frame_ready_rest)
                        vg.graph_front.replay()
                        vg.front_done.record(fs)
            with torch.cuda.stream(side):
                for v, vg in enumerate(self.views):
                    if self.front_stream:
                        side.wait_event(vg.front_done)
                    if v == 1 and rest is not None and not self.front_stream:
                        side.wait_event(self._frame_ready_rest)     # (the other views' old id lists have been read)
                    vg.graph.replay()
                    if fx is not None:
                        fx.static_begin_view(v, vg.out["union_mask"])   # (RCCL orders the mask's all-reduce behind this stream)
                    vg.fwd_done.record(side)
            for bs in self.extra_bwd_streams[:nb - 1]:
                bs.wait_event(self._frame_ready if rest is None else self._frame_ready_rest)
        for v, vg in enumerate(self.views):
            # ---- forward (already on its way in the overlapped form)
            if side is None:
                if vg.graph_front is not None:
                    vg.graph_front.replay()
                vg.graph.replay()
                if fx is not None:
                    fx.static_begin_view(v, vg.out["union_mask"])
            # ---- image half of the backward: needs nothing from the other ranks
            bs = main if v % nb == 0 else self.extra_bwd_streams[v % nb - 1]
            if side is not None:
                bs.wait_event(vg.fwd_done)
            with torch.cuda.stream(bs):
                vg.graph_bwd.replay()
                if tail is not None:
                    vg.bwd_done.record(bs)
            # ---- Gaussian half: the only writer of the per-Gaussian gradient rows, one view after the other
            ts = main if tail is None else tail
            if tail is not None:
                tail.wait_event(vg.bwd_done)
            with torch.cuda.stream(ts):
                if fx is not None:
                    fx.static_targets(v)    # waits for the union mask; slot map + cleared buffer rows (two launches)
                if vg.graph_tail is not None:
                    vg.graph_tail.replay()
                if fx is not None:
                    fx.static_end_view(v)   # all-reduce of view v's rows (async); reduced rows of view v - 1 added to the dense buffer
                if v == self.V - 1 and fx is not None:
                    fx.static_end_frame()
                vg.done.record(ts)
        if side is not None and self.front_stream:
            main.wait_stream(self.front_stream_h)
        if tail is not None:
            main.wait_event(self.views[-1].done)     # the frame's gradients are complete for whatever the caller enqueues next
            for bs in self.extra_bwd_streams[:nb - 1]:
                main.wait_stream(bs)
        self._sum_pose_slots()

    def _sum_pose_slots(self) -> None:
        """Camera-pose gradients of all views: ONE reduction of the views' slots per frame (``viewmat.grad`` of every camera is a row of
        the result) instead of one per view."""
        if self._defer_pose:
            torch.sum(self._tails[:, :L.POSE_GRAD_SLOTS].view(self.V, L.POSE_GRAD_SLOTS, 4, 4), dim=1, out=self._vm)

    def _step_phase_shifted(self, main, rest) -> None:
        """Three streams, software-pipelined on the host so that every event is recorded before it is waited for: iteration k enqueues
        front(k) [after forward(k - 2)], image(k) = compositor + colour transform + loss [after front(k) and after the compositor
        backward of view k - 2], and the backward of view k - 1."""
        side, fs = self.side_stream, self.front_stream_h
        self._frame_ready.record(main)
        side.wait_event(self._frame_ready)
        fs.wait_event(self._frame_ready)
        if rest is not None:
            rest.replay()
            self._frame_ready_rest.record(main)
        V = self.V
        for k in range(V + 1):
            if k < V:
                vg = self.views[k]
                with torch.cuda.stream(fs):
                    if k == 1 and rest is not None:
                        fs.wait_event(self._frame_ready_rest)
                    if k >= 2:
                        fs.wait_event(self.views[k - 2].fwd_done)
                    vg.graph_front.replay()
                    vg.front_done.record(fs)
                with torch.cuda.stream(side):
                    side.wait_event(vg.front_done)
                    if k >= 2:
                        side.wait_event(self.views[k - 2].bwd_done)
                    vg.graph.replay()
                    vg.fwd_done.record(side)
            if k >= 1:
                vb = self.views[k - 1]
                main.wait_event(vb.fwd_done)
                vb.graph_bwd.replay()
                vb.bwd_done.record(main)
                vb.graph_tail.replay()
                vb.done.record(main)
        main.wait_stream(fs)
        main.wait_stream(side)
        self._sum_pose_slots()

    def mark_samples(self, name: str):
        """Milliseconds of every timing mark pair ``name`` captured into the view graphs (``_lib.enable_timers`` on during the
        capture), as recorded by the last replay; synchronise first."""
        return L.graph_mark_samples(name, getattr(self, "marks", {}))

    # ---- after the fact ----------------------------------------------------------------------------------------------------------
    def counts(self):
        """[(M, visible)] per view as last written by the GPU (synchronise first for the current frame's values)."""
        return [c.observed() for c in self.caps]

    def _check_counts(self, raise_on_overflow: bool = False) -> bool:
        ok = True
        for v, c in enumerate(self.caps):
            M, n_vis = c.observed()
            if c.overflowed() or M > c.m_cap or n_vis > c.nvis_cap:
                if raise_on_overflow:
                    raise L.BdsError(f"view {v}: list counts (M = {M}, visible = {n_vis}) exceed the calibrated capacities "
                                     f"({c.m_cap}, {c.nvis_cap}) right after calibration")
                ok = False
                self._grow(v, M, n_vis)
        if self.fx is not None:      # the ranks' unions (identical counts on every rank: every rank decides alike)
            for v, n in enumerate(self.fx.static_counts()):
                self._unions[v] = max(self._unions[v], n)
                if n > self.fx.cap:
                    if raise_on_overflow:
                        raise L.BdsError(f"view {v}: union of the ranks' visible sets ({n}) exceeds the exchange capacity ({self.fx.cap})")
                    ok = False
        return ok

    def valid(self) -> bool:
        """Wait for the frame in flight; True if every view's lists (and exchange buffers) fitted.  Otherwise the capacities are grown,
        the graphs captured again and False is returned: the frame's gradients are incomplete (the overflowing view rendered
        nothing) -- repeat it."""
        for vg in self.views:
            vg.done.synchronize()
        if self._check_counts():
            # keep ahead of a growing scene: re-provision when a count comes within 8 % of its capacity
            grow = [v for v, c in enumerate(self.caps) if c.observed()[0] > 0.92 * c.m_cap or (c.observed()[1] > 0.92 * c.nvis_cap and c.nvis_cap < self.N)]
            near = self.fx is not None and max(self._unions) > 0.92 * self.fx.cap and self.fx.cap < (self.N + 3) // 4 * 4
            if grow or near:
                for v in grow:
                    self._grow(v, *self.caps[v].observed())
                self.capture()
            return True
        self.capture()
        return False
