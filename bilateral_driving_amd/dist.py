"""Multi-GPU data parallelism over VIEWS (new capability: the reference trains one view per step on
one GPU -- /root/reference/project/models/trainers/base.py:411, scripts/train.sh:2,23).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  Every rank holds a full replica of the Gaussian parameters and renders its own view; the only
exchange step of an iteration is ONE sum-all-reduce of the flat per-Gaussian gradient buffer
(59 floats/Gaussian at SH degree 3: 3 means + 4 quats + 3 scales + 1 opacity + 48 SH) plus the small
bilateral-grid / pose tail.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

# Collectives at world size 1.  A one-rank all-reduce is the identity, so every guard below used to skip it -- and the RCCL call
# sequence (uint8 MAX masks, async work handles next to the two-stream graph replay, captures next to RCCL's proxy thread) had never
# executed on a one-GPU box.  ``force_collectives(True)`` (or BDS_FORCE_COLLECTIVES=1) makes an initialised process group of ONE rank
# behave like a group of several: every collective of the exchange is really issued.  Gradients must then equal the no-exchange frame.
_FORCE: Optional[bool] = None
ISSUED = {"all_reduce": 0, "broadcast": 0, "bytes": 0}      # collectives this process has issued through this module


def force_collectives(on: Optional[bool]) -> None:
    """True / False: override; None: follow the environment (BDS_FORCE_COLLECTIVES=1)."""
    global _FORCE
    _FORCE = on


def collectives_forced() -> bool:
    import os
    return bool(_FORCE) if _FORCE is not None else os.environ.get("BDS_FORCE_COLLECTIVES", "0") == "1"


def _all_reduce(t: Tensor, op, async_op: bool = False):
    ISSUED["all_reduce"] += 1
    ISSUED["bytes"] += t.numel() * t.element_size()
    return dist.all_reduce(t, op=op, async_op=async_op)


class FlatGradients:
    """One contiguous fp32 communication buffer for the gradients of a fixed parameter list.

    Single GPU: nothing is copied -- ``zero()`` drops the old gradients and autograd adopts the kernels'
    gradient outputs as ``param.grad`` without an accumulation pass.  Multi GPU: ``all_reduce()`` packs the
    gradients into the flat buffer (one read + one write of the payload), runs ONE sum-all-reduce, and
    re-points every ``param.grad`` at its slice of the reduced buffer."""

    def __init__(self, params: Iterable[Tensor], sparse_rows: bool = False, row_block: bool = False):
        """``row_block``: the gradients of the first four parameters -- means [N,3], quats [N,4], log_scales [N,3], opacity logits [N]
        (``ROW_NAMES`` order) -- are the COLUMNS of one [N,16] block at the head of the flat buffer (64-byte rows {mean 3, logit |
        quat 4 | log_scale 3, - | - - - -}): a list-driven backward then updates ONE line per visible Gaussian instead of four
        partly used ones (csrc/bds_common.h GradLayout: recognised by the addresses).  Their ``.grad`` are strided views; the five
        unused columns stay zero.  Single-process use (``graph_view.FrameGraph`` without an exchange).
        ``sparse_rows``: the flat buffer is kept all-zero between steps by clearing only the rows that were written
        (``mark_list`` / ``begin_rows_union`` tell which), so that a producer may touch just the rows it needs
        (``fused_view(grad_arena=..., arena_rows=1 | 2)``: a view sees ~15 % of the Gaussians).  Whenever the book is incomplete
        the whole buffer is cleared."""
        self.sparse_rows = bool(sparse_rows)
        self._dirty: Optional[List[Tensor]] = None   # id lists (int32) of the rows that may be non-zero (None: unknown -> dense clear)
        self._clean = False                          # the flat buffer is known to be all zeros
        self._arena_names: List[str] = []
        self.params: List[Tensor] = [p for p in params]
        assert self.params, "no parameters"
        dev = self.params[0].device
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev and p.requires_grad
        self.row_block = bool(row_block)
        if self.row_block:
            N = self.params[0].shape[0]
            assert len(self.params) >= 4 and [tuple(p.shape) for p in self.params[:4]] == [(N, 3), (N, 4), (N, 3), (N,)], \
                "row_block: means [N,3], quats [N,4], log_scales [N,3], opacity logits [N] first"
        self.total = sum(p.numel() for p in self.params) + (5 * self.params[0].shape[0] if self.row_block else 0)
        self._flat: Optional[Tensor] = None
        self._views: List[Tensor] = []
        self._work = None
        self._union: Optional[Tensor] = None     # rows (first-dim entries) touched on ANY rank this step
        self._union_work = None
        self.last_payload_bytes = 0

    @property
    def flat(self) -> Tensor:
        if self._flat is None:
            self._flat = torch.zeros(self.total, device=self.params[0].device, dtype=torch.float32)
            off, rest = 0, self.params
            if self.row_block:
                N = self.params[0].shape[0]
                block = self._flat[:N * 16].view(N, 16)
                self._views += [block[:, 0:3], block[:, 4:8], block[:, 8:11], block[:, 3]]
                off, rest = N * 16, self.params[4:]
            for p in rest:
                self._views.append(self._flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return self._flat

    def zero(self) -> None:
        for p in self.params:
            p.grad = None
        if self._union_work is not None:      # a union that no exchange consumed: drop it (and distrust the book)
            self._union_work.wait()
            self._union, self._union_work = None, None
            self._dirty = None
        if self.sparse_rows:
            self._clear_rows()

    def mark_list(self, ids: Tensor) -> None:
        """Rows (first-dim entries, as an int32 id list, e.g. ``info["visible_ids"]`` of a fused view) that a backward of THIS
        process may write; call once per view, the lists of a step accumulate (with several ranks ``begin_rows_union`` records
        the union over the ranks instead)."""
        if not self.sparse_rows:
            return
        if not self._clean and self._dirty is None:
            return                                   # state unknown (never cleared yet): stays unknown -> dense clear next time
        self._dirty = (self._dirty or []) + [ids.reshape(-1).to(torch.int32)]
        self._clean = False

    def mark_rows(self, touched: Tensor) -> None:
        """The same from a boolean / uint8 mask [N] (costs a host sync: prefer ``mark_list``)."""
        if self.sparse_rows:
            self.mark_list(touched.reshape(-1).nonzero().squeeze(1))

    @property
    def rows_clean(self) -> bool:
        """True right after ``zero()`` in sparse_rows mode: every row of the arena is zero."""
        return self.sparse_rows and self._clean

    def _clear_rows(self) -> None:
        from . import _lib as L   # the row-wise clear is a libbds kernel
        flat = self.flat
        row = dict(zip(self._arena_names, self._views))
        keys = ("means", "quats", "log_scales", "opacity_logits", "sh")
        if self._dirty is not None and flat.is_cuda and all(k in row for k in keys) and row["sh"].dim() == 3:
            K = row["sh"].shape[1]
            # (row_block: the four small gradients are strided columns of one [N,16] block -- the kernel recognises the layout by the
            #  addresses, csrc/bds_common.h GradLayout -- so their base addresses go in as they are)
            addr = (lambda t: t.data_ptr()) if self.row_block else L.ptr
            for ids in self._dirty:
                L.check(L.lib().bds_view_grads_clear_list(ids.numel(), L.ptr(ids.contiguous()), K, addr(row["means"]), addr(row["quats"]),
                                                          addr(row["log_scales"]), addr(row["opacity_logits"]), L.ptr(row["sh"]),
                                                          L.stream()), "bds_view_grads_clear_list")
            # the other slices are dense: pack() / autograd overwrite (or zero) them before they are read
        elif self._dirty is not None and not flat.is_cuda:   # CPU tensors (the gloo tests of the exchange logic): same effect
            n_rows = max((v.shape[0] for v in self._views if v.dim() >= 1), default=0)
            for ids in self._dirty:
                il = ids[ids >= 0].long()
                for v in self._views:
                    if v.dim() >= 1 and v.shape[0] == n_rows:
                        v.index_fill_(0, il, 0.0)
        else:
            flat.zero_()
        self._dirty = []
        self._clean = True

    def arena(self, names: Iterable[str]) -> Dict[str, Tensor]:
        """name -> slice of the flat buffer, for producers that can write a parameter's gradient in place
        (fused_view(grad_arena=...)); ``pack()`` then finds the gradient already where it belongs."""
        _ = self.flat
        self._arena_names = list(names)
        return {n: v for n, v in zip(names, self._views)}

    def pack(self) -> Tensor:
        flat = self.flat
        for p, v in zip(self.params, self._views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
        return flat

    def begin_rows_union(self, touched: Tensor) -> None:
        """Optional, call right after the forward pass: ``touched`` [N] marks the rows (Gaussians) that can receive a
        non-zero gradient on this rank (radii > 0 -- every backward kernel writes exact zeros for culled Gaussians).
        The element-wise OR over the ranks is started asynchronously; the next ``all_reduce()`` then exchanges only the
        rows touched on at least one rank (a view sees ~15 % of the Gaussians, so with few ranks most of the 59 floats per
        Gaussian that a dense all-reduce would move are zeros on every rank).  The result is identical to the dense sum."""
        if not _active():
            return
        self._union = touched.reshape(-1).to(torch.uint8).clone()
        self._union_work = _all_reduce(self._union, dist.ReduceOp.MAX, async_op=True)
        self._clean = False

    def _all_reduce_rows(self, flat: Tensor) -> bool:
        """Exchange only the rows of the union.  Returns False (nothing done) when that would not pay."""
        self._union_work.wait()
        union, self._union, self._union_work = self._union, None, None
        n_rows = union.numel()
        idx = union.nonzero().squeeze(1)            # same on every rank (host sync: the exchange waits for backward anyway)
        if self.sparse_rows:
            self._dirty = [idx.to(torch.int32)]     # after the exchange exactly the rows of the union may be non-zero (also on the dense path)
        if idx.numel() > 0.85 * n_rows:
            return False
        row_views = [v for v in self._views if v.dim() >= 1 and v.shape[0] == n_rows]
        other_views = [v for v in self._views if not (v.dim() >= 1 and v.shape[0] == n_rows)]
        if not row_views:
            return False
        k = idx.numel()
        parts = [v.reshape(n_rows, -1).index_select(0, idx).reshape(-1) for v in row_views] + [v.reshape(-1) for v in other_views]
        comm = torch.cat(parts)
        _all_reduce(comm, dist.ReduceOp.SUM)
        off = 0
        for v in row_views:
            w = v.reshape(n_rows, -1)
            w.index_copy_(0, idx, comm[off:off + k * w.shape[1]].view(k, w.shape[1]))
            off += k * w.shape[1]
        for v in other_views:
            v.copy_(comm[off:off + v.numel()].view_as(v))
            off += v.numel()
        self.last_payload_bytes = comm.numel() * 4
        return True

    def all_reduce(self, average: bool = False, async_op: bool = False):
        """Sum (or average) the gradients over all ranks.  No-op for world size 1."""
        if not _active():
            return None
        # (row_block is the single-process layout: the exchange's compact buffers and pack() know the five arrays only)
        assert not self.row_block, "FlatGradients(row_block=True) can not be exchanged over ranks"
        flat = self.pack()
        for p, v in zip(self.params, self._views):
            p.grad = v
        if self._union_work is not None and self._all_reduce_rows(flat):
            if average:
                flat.div_(dist.get_world_size())
            return None
        if self.sparse_rows:
            # dense exchange without a union (the caller only used mark_rows): rows touched on OTHER ranks are non-zero now,
            # and this rank's book does not know them -> the next zero() has to clear everything
            self._dirty = None
            self._clean = False
        self.last_payload_bytes = flat.numel() * 4
        self._work = _all_reduce(flat, dist.ReduceOp.SUM, async_op=async_op)
        if not async_op and average:
            flat.div_(dist.get_world_size())
        return self._work

    def wait(self, average: bool = False) -> None:
        if self._work is not None:
            self._work.wait()
            self._work = None
            if average:
                self.flat.div_(dist.get_world_size())

    @property
    def nbytes(self) -> int:
        return self.total * 4


ROW_NAMES = ("means", "quats", "log_scales", "opacity_logits", "sh")   # the per-Gaussian parameters of the fused view


class FrameExchange:
    """One training step of view-parallel data parallelism = one FRAME: every rank renders the V views of its timestep
    (BASELINE.json: "6 cams x 8 timesteps sharded across 8 GPUs"), their gradients are summed over views and ranks.

    The exchange is made per VIEW and overlapped with the next view's compute.  A view sees ~15 % of the Gaussians and every
    backward kernel writes only those rows, so instead of ONE dense all-reduce of 59 floats x N at the end of the frame (472 MB
    at 2 M Gaussians: ~2.6 ms over xGMI behind 7.5 ms of compute, nothing to overlap it with) each view does:

      1. right after its forward pass: an asynchronous MAX-all-reduce of the uint8 visibility mask (N bytes) -> union over the ranks;
      2. inside its backward, before the list-driven kernels: slot map of the union (prefix sum) -> the kernels store the rows of
         the visible Gaussians at their union slot of a zeroed COMPACT buffer [capacity, 59] (``fused_view(grad_sink=self)``);
      3. after its backward: asynchronous SUM-all-reduce of the compact buffer (RCCL runs it on its own stream while the next
         view's kernels run on the compute stream);
      4. one view later: the reduced rows are added to the dense gradient buffer (``param.grad``) through the union's id list.

    Only the last view's exchange is exposed.  Capacity = ``headroom`` x the LARGEST union seen over all views so far and never
    shrinks: during the first frame every view's union count is read back (one host wait per view, once) and the buffers grow on
    demand; afterwards the counts travel back asynchronously and ``end_frame`` -- which has waited for the last exchange anyway --
    checks them BEFORE the caller's optimizer step: a union that outgrew the capacity raises there (the frame's gradients are
    incomplete, nothing has consumed them yet), one that came within 8 % of it re-sizes the buffers for the next frame.
    World size 1: ``view_kwargs`` selects the in-place arena modes of ``fused_view`` and nothing is exchanged.  ``force=True`` runs the
    compact path without any collective (single-GPU test of the kernels the exchange depends on)."""

    def __init__(self, flat: FlatGradients, names: Iterable[str], headroom: float = 1.25, n_buffers: int = 3, force: bool = False,
                 per_view: bool = True):
        self.flat = flat
        self.names = list(names)
        assert flat.sparse_rows, "FrameExchange keeps the dense buffer clean row-wise: FlatGradients(sparse_rows=True)"
        assert not flat.row_block, "FrameExchange: FlatGradients(row_block=True) is the single-process layout"
        self.arena = flat.arena(self.names)
        assert sorted(self.names[:len(ROW_NAMES)]) == sorted(ROW_NAMES), "the five per-Gaussian parameters must lead the parameter list"
        self.N = self.arena["means"].shape[0]
        self.K = self.arena["sh"].shape[1]
        self.row_floats = 3 + 4 + 3 + 1 + self.K * 3
        self.world = dist.get_world_size() if _active() else 1
        self.coll = _active()               # collectives are issued (world > 1, or one rank with ``force_collectives``)
        # per_view=False: the frame's gradients accumulate in place exactly as at world size 1 and ``end_frame`` sums the dense buffer over
        # the ranks ONCE (fewer bytes than the per-view exchanges when the ranks' unions approach the whole scene, none of them hidden:
        # ``plan_exchange`` prices the two)
        self.per_view = bool(per_view)
        self.frame_reduce = self.coll and not self.per_view and not force
        self.active = bool(force) or (self.coll and self.per_view)
        self.headroom, self.n_buffers = float(headroom), int(n_buffers)
        n_row = sum(v.numel() for v in flat._views[:len(ROW_NAMES)])
        self._tail = flat.flat[n_row:] if (not self.active and flat.total > n_row) else None
        self.cap = 0
        self._max_union = 0                 # largest union count of any view so far (the capacity follows it, never shrinks)
        self._sized = False                 # the first frame reads every view's count back; later frames check at end_frame
        self._bufs: List[Tensor] = []
        self._free: List[int] = []
        self._pending: List[tuple] = []     # (work, buffer index, ids) of exchanges in flight, oldest first
        self._union = None                  # (mask uint8 [N], work) of the current view
        self._cur = None                    # (buffer index, ids) handed out by targets() for the current view
        self._counts = []                   # (pinned int64 tensor, event, capacity) for the deferred overflow check
        self.payload_bytes = 0              # bytes all-reduced during the last frame (per rank)
        self.n_exchanges = 0

    # ---- per frame -------------------------------------------------------------------------------------------------
    def begin_frame(self) -> None:
        self._check_overflow()              # (counts of a frame whose end_frame was skipped)
        if self.active and self._bufs and self._max_union > 0.92 * self.cap and self.cap < (self.N + 3) // 4 * 4 and not self._pending:
            self._allocate(self._wanted_cap())   # a union came within 8 % of the capacity: re-size upward between frames
        self.flat.zero()
        if self.active:   # the reduced rows are added into the dense buffer, which IS the parameters' .grad; the small dense tail
            #               (grids, ...) is accumulated by autograd as usual and packed in end_frame
            for p, v in zip(self.flat.params[:len(ROW_NAMES)], self.flat._views[:len(ROW_NAMES)]):
                p.grad = v
        elif self._tail is not None:
            # world size 1: the small dense tail (grids, ...) is accumulated IN PLACE by the producing kernels (fused_view's grid
            # gradients, the TV term of the loss: ``tail_grads``) -- one fill per frame here instead of an autograd sum + an
            # accumulation pass per tensor and view
            self._tail.zero_()
            for p, v in zip(self.flat.params[len(ROW_NAMES):], self.flat._views[len(ROW_NAMES):]):
                p.grad = v
        self.payload_bytes, self.n_exchanges = 0, 0

    def view_kwargs(self, v: int) -> dict:
        """Keyword arguments of ``fused_view`` / ``harness.render_view`` for the v-th view of the frame."""
        if self.active:
            return dict(grad_sink=self)
        return dict(grad_arena=self.arena, arena_rows=1 if v == 0 else 2)

    def tail_grads(self, prefix: str = "grid") -> Optional[List[Tensor]]:
        """World size 1: the ``.grad`` slices of the parameters named ``prefix + i`` (zeroed by ``begin_frame``), for producers
        that add their gradient in place (``losses.photometric_tv_loss(grid_grads=...)``); None while an exchange is active
        (autograd accumulates the tail then)."""
        if self.active or self._tail is None:
            return None
        return [self.arena[n] for n in self.names if n.startswith(prefix)]

    def begin_view(self, info) -> None:
        """Right after the view's forward pass (``info`` = its info dict)."""
        if not self.active:
            self.flat.mark_list(info["visible_ids"])
            return
        mask = (info["radii"].reshape(-1) > 0).to(torch.uint8)
        work = _all_reduce(mask, dist.ReduceOp.MAX, async_op=True) if self.coll else None
        self._union = (mask, work)

    def targets(self, visible_ids: Tensor):
        """Called by the fused view's backward: compact destination buffers + the slot map of the ranks' union."""
        mask, work = self._union
        self._union = None
        if work is not None:
            work.wait()
        if mask.is_cuda and self._sized:
            return self._targets_device(mask)
        slot = torch.cumsum(mask, 0, dtype=torch.int32) - 1
        if not self._sized:   # first frame: every view's union is read back (one host wait per view, once) and the buffers grow on demand
            n = int(slot[-1]) + 1
            self._max_union = max(self._max_union, n)
            if n > self.cap:
                while self._pending:        # buffers in flight have the old size: drain them first
                    self._retire()
                self._allocate(self._wanted_cap())
        cnt = torch.empty(1, dtype=torch.int64).pin_memory() if mask.is_cuda else torch.empty(1, dtype=torch.int64)
        cnt.copy_(slot[-1:] + 1, non_blocking=True)
        ev = None
        if mask.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._counts.append((cnt, ev, self.cap))
        row_map = slot.clamp_(max=self.cap - 1)
        # ids[s] = the Gaussian at union slot s, -1 beyond the union (no host sync: scatter through a dump slot)
        if getattr(self, "_iota", None) is None or self._iota.numel() != self.N or self._iota.device != mask.device:
            self._iota = torch.arange(self.N, device=mask.device, dtype=torch.int32)
        ids = torch.full((self.cap + 1,), -1, device=mask.device, dtype=torch.int32)
        ids.scatter_(0, torch.where(mask > 0, row_map, self.cap).long(), self._iota)
        ids = ids[:self.cap]
        if not self._free:      # every buffer is in flight: retire the oldest exchange first
            self._retire()
        b = self._free.pop()
        buf = self._bufs[b]
        buf.zero_()
        self._cur = (b, ids)
        return self._views_of(buf), row_map

    def _targets_device(self, mask: Tensor):
        """``targets`` on the GPU once the buffers are sized: mask -> slot map, id list, cleared buffer rows and the count in TWO
        launches of libbds (bds_union_slots) instead of ten framework operators on N-element tensors."""
        from . import _lib as L
        lib = L.lib()
        dev = mask.device
        if not self._bufs:          # (sized by the static form, whose buffers are per view: the rotating ones are made on first use)
            self._allocate(max(self.cap, self._wanted_cap()))
        if not self._free:
            self._retire()
        b = self._free.pop()
        buf = self._bufs[b]
        assert buf.numel() == self.cap * self.row_floats, "exchange buffers and capacity out of step"
        views = self._views_of(buf)
        # slot map, id list, count word and workspace: one set per rotating buffer, made with it (a pinned allocation and a device
        # pointer look-up per view would undo what the two-launch form saves); the count words stay alive with the exchange
        row_map, ids, cnt_all, ws = self._scratch[b]
        wsb = ws.numel()
        k = self._uses[b] = (self._uses[b] + 1) % cnt_all.numel()     # (a buffer serves several views of a frame: one count word each)
        cnt = cnt_all[k:k + 1]
        L.check(lib.bds_union_slots(self.N, L.ptr(mask), self.cap, self.K, L.ptr(row_map), L.ptr(ids), L.ptr(views["means"]),
                                    L.ptr(views["quats"]), L.ptr(views["log_scales"]), L.ptr(views["opacity_logits"]), L.ptr(views["sh"]),
                                    L.ptr(ws), wsb, None, cnt.data_ptr(), L.stream()), "bds_union_slots")
        ev = torch.cuda.Event()
        ev.record()
        self._counts.append((cnt, ev, self.cap))
        self._cur = (b, ids)
        return views, row_map

    # ---- static form (graph_view.FrameGraph at world size > 1): per-view buffers at fixed addresses ---------------------------------
    def static_setup(self, union_counts: Iterable[int]) -> None:
        """One compact buffer, slot map, id list, mask and count word PER VIEW of the frame, at addresses that captured hipGraphs can
        hold; capacity from ``union_counts`` (the views' union sizes of a calibration visit: identical on every rank)."""
        from . import _lib as L
        counts = [int(c) for c in union_counts]
        self._max_union = max([self._max_union] + counts)
        while self._pending:
            self._retire()
        cap = self._wanted_cap()
        if cap != self.cap:       # the eager path's rotating buffers (and their per-buffer scratch) have the old size: made again on use
            self._bufs, self._free, self._scratch, self._uses = [], [], [], []
        self.cap = cap
        dev = self.arena["means"].device
        V = len(counts)
        self._sbuf = [torch.zeros(self.cap * self.row_floats, device=dev, dtype=torch.float32) for _ in range(V)]
        self._srow = [torch.zeros(self.N, device=dev, dtype=torch.int32) for _ in range(V)]
        self._sids = [torch.full((self.cap,), -1, device=dev, dtype=torch.int32) for _ in range(V)]
        self._scnt = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(V)]
        wsb = L.lib().bds_union_slots_workspace_bytes(self.N)
        self._sws = [torch.empty(wsb, device=dev, dtype=torch.uint8) for _ in range(V)]
        self._swork = [None] * V
        self._spending: List[tuple] = []
        self._sized = True
        self._static = True

    def static_sink(self, v: int):
        """The ``grad_sink`` a captured view hands to its backward: ``targets`` returns view v's fixed buffers and slot map (filled by
        ``static_targets`` right before the Gaussian half is replayed)."""
        fx = self

        class _Sink:
            def targets(self, visible_ids):
                return fx._views_of(fx._sbuf[v]), fx._srow[v]
        return _Sink()

    def static_begin_view(self, v: int, mask: Tensor) -> None:
        """After view v's forward: start the MAX-all-reduce of its visibility mask (uint8 [N], reduced in place)."""
        self._smask = getattr(self, "_smask", {})
        self._smask[v] = mask
        self._swork[v] = _all_reduce(mask, dist.ReduceOp.MAX, async_op=True) if self.coll else None

    def static_targets(self, v: int) -> None:
        """Before view v's Gaussian half: union mask -> slot map, id list, cleared buffer rows, count (two launches)."""
        from . import _lib as L
        if self._swork[v] is not None:
            self._swork[v].wait()
            self._swork[v] = None
        views = self._views_of(self._sbuf[v])
        lib = L.lib()
        L.check(lib.bds_union_slots(self.N, L.ptr(self._smask[v]), self.cap, self.K, L.ptr(self._srow[v]), L.ptr(self._sids[v]),
                                    L.ptr(views["means"]), L.ptr(views["quats"]), L.ptr(views["log_scales"]), L.ptr(views["opacity_logits"]),
                                    L.ptr(views["sh"]), L.ptr(self._sws[v]), self._sws[v].numel(), None, self._scnt[v].data_ptr(), L.stream()),
                "bds_union_slots")

    def static_end_view(self, v: int) -> None:
        """After view v's Gaussian half: start the SUM-all-reduce of its compact buffer; add the PREVIOUS view's reduced rows."""
        buf = self._sbuf[v]
        work = _all_reduce(buf, dist.ReduceOp.SUM, async_op=True) if self.coll else None
        self.payload_bytes += buf.numel() * 4
        self.n_exchanges += 1
        self._spending.append((work, v))
        while len(self._spending) > 1:
            self._static_retire()

    def _static_retire(self) -> None:
        from . import _lib as L
        work, v = self._spending.pop(0)
        if work is not None:
            work.wait()
        src, dst, ids = self._views_of(self._sbuf[v]), self.arena, self._sids[v]
        L.check(L.lib().bds_view_grads_add_list(ids.numel(), L.ptr(ids), self.K, L.ptr(src["means"]), L.ptr(src["quats"]),
                                                L.ptr(src["log_scales"]), L.ptr(src["opacity_logits"]), L.ptr(src["sh"]),
                                                L.ptr(dst["means"]), L.ptr(dst["quats"]), L.ptr(dst["log_scales"]),
                                                L.ptr(dst["opacity_logits"]), L.ptr(dst["sh"]), L.stream()), "bds_view_grads_add_list")
        self.flat.mark_list(ids)

    def static_end_frame(self) -> None:
        """Drain the exchanges and sum the dense tail (grids, ...), which the views accumulated in place, over the ranks."""
        while self._spending:
            self._static_retire()
        n_row = self.N * self.row_floats
        tail = self.flat.flat[n_row:]
        if tail.numel() and self.coll:
            _all_reduce(tail, dist.ReduceOp.SUM)
            self.payload_bytes += tail.numel() * 4

    def static_counts(self):
        """Union sizes of the last frame's views (synchronise first); a value above ``cap`` = that view's exchange overflowed."""
        return [int(c[0]) for c in self._scnt]

    def end_view(self) -> None:
        """Right after the view's backward pass."""
        if not self.active:
            return
        b, ids = self._cur
        self._cur = None
        buf = self._bufs[b]
        work = _all_reduce(buf, dist.ReduceOp.SUM, async_op=True) if self.coll else None
        self.payload_bytes += buf.numel() * 4
        self.n_exchanges += 1
        self._pending.append((work, b, ids))
        while len(self._pending) > 1:      # the previous view's exchange ran behind this view's compute
            self._retire()

    def end_frame(self) -> None:
        """After the last view: drain the exchanges, check the frame's union counts (raises BEFORE the caller's optimizer step if one
        outgrew the capacity) and sum the small dense tail (grids, ...) over the ranks.  ``per_view=False``: the one dense all-reduce."""
        if self.frame_reduce:
            self.reduce_frame()
            return
        if not self.active:
            return
        while self._pending:
            self._retire()
        self._sized = True
        self._check_overflow()
        n_row = self.N * self.row_floats
        tail = self.flat.flat[n_row:]
        if tail.numel():
            for p, v in zip(self.flat.params[len(ROW_NAMES):], self.flat._views[len(ROW_NAMES):]):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                p.grad = v
            if self.coll:
                _all_reduce(tail, dist.ReduceOp.SUM)
                self.payload_bytes += tail.numel() * 4

    def reduce_frame(self) -> None:
        """``per_view=False``: sum the whole flat gradient buffer over the ranks (on the current stream; RCCL orders it behind the frame's
        kernels).  Rows other ranks wrote are non-zero afterwards: the next ``begin_frame`` clears the buffer densely."""
        assert self.frame_reduce
        _all_reduce(self.flat.flat, dist.ReduceOp.SUM)
        self.flat._dirty, self.flat._clean = None, False
        self.payload_bytes, self.n_exchanges = self.flat.nbytes, 1

    # ---- internals -------------------------------------------------------------------------------------------------
    def _wanted_cap(self) -> int:
        cap = int(self._max_union * self.headroom) + 64
        return max(self.cap, min((cap + 3) // 4 * 4, (self.N + 3) // 4 * 4))   # multiple of 4: every sub-array starts 16-byte aligned

    def _allocate(self, cap: int) -> None:
        assert not self._pending and cap >= self.cap
        self.cap = cap
        dev = self.arena["means"].device
        self._bufs = [torch.zeros(self.cap * self.row_floats, device=dev, dtype=torch.float32) for _ in range(self.n_buffers)]
        self._free = list(range(self.n_buffers))
        self._scratch = []
        if dev.type == "cuda":
            from . import _lib as L
            wsb = L.lib().bds_union_slots_workspace_bytes(self.N)
            self._scratch = [(torch.empty(self.N, device=dev, dtype=torch.int32), torch.empty(self.cap, device=dev, dtype=torch.int32),
                              torch.zeros(64, dtype=torch.int64).pin_memory(), torch.empty(wsb, device=dev, dtype=torch.uint8))
                             for _ in range(self.n_buffers)]
        self._uses = [0] * self.n_buffers

    def _views_of(self, buf: Tensor) -> Dict[str, Tensor]:
        c, K = self.cap, self.K
        o, out = 0, {}
        for name, shape in (("means", (c, 3)), ("quats", (c, 4)), ("log_scales", (c, 3)), ("opacity_logits", (c,)), ("sh", (c, K, 3))):
            n = 1
            for d in shape:
                n *= d
            out[name] = buf[o:o + n].view(shape)
            o += n
        return out

    def _retire(self) -> None:
        work, b, ids = self._pending.pop(0)
        if work is not None:
            work.wait()
        src, dst = self._views_of(self._bufs[b]), self.arena
        if src["means"].is_cuda:
            from . import _lib as L
            L.check(L.lib().bds_view_grads_add_list(ids.numel(), L.ptr(ids), self.K, L.ptr(src["means"]), L.ptr(src["quats"]),
                                                    L.ptr(src["log_scales"]), L.ptr(src["opacity_logits"]), L.ptr(src["sh"]),
                                                    L.ptr(dst["means"]), L.ptr(dst["quats"]), L.ptr(dst["log_scales"]),
                                                    L.ptr(dst["opacity_logits"]), L.ptr(dst["sh"]), L.stream()), "bds_view_grads_add_list")
        else:   # CPU tensors (gloo tests of the exchange logic)
            ok = ids >= 0
            il = ids[ok].long()
            for k in ROW_NAMES:
                dst[k].index_add_(0, il, src[k][ok])
        # a COPY of the list: ``ids`` is buffer b's scratch, and b goes back to the free list now -- a later view of this same frame may
        # take it and overwrite the list before the next begin_frame clears the rows it names (found by the third eager frame of
        # tests/rccl_world1_worker.py: V = n_buffers = 3, view 2 re-used view 0's buffer and view 0's rows were never cleared)
        self.flat.mark_list(ids.clone())
        self._free.append(b)

    def _check_overflow(self) -> None:
        """Union counts recorded since the last check (every rank sees the same counts and decides alike)."""
        counts, self._counts = self._counts, []
        worst = None
        for cnt, ev, cap in counts:
            if ev is not None:
                ev.synchronize()    # (never query(): every rank must decide alike)
            n = int(cnt[0])
            self._max_union = max(self._max_union, n)    # the capacity follows the largest union: begin_frame re-sizes upward
            if n > cap and (worst is None or n > worst[0]):
                worst = (n, cap)
        if worst is not None:
            raise RuntimeError(f"FrameExchange: the union of the ranks' visible sets ({worst[0]} Gaussians) outgrew the exchange "
                               f"capacity ({worst[1]}): this frame's gradients are incomplete -- discard them and repeat the frame "
                               f"(the buffers are re-sized at the next begin_frame)")


def reduce_densify_stats(grad_norm_accum: Tensor, vis_counts: Tensor, max_2d_size: Tensor) -> None:
    """Per-view densification statistics (models/gaussians/vanilla.py:163-191) become global:
    sums for the accumulated gradient norm / visibility counts, max for the screen-space size."""
    if not _active():
        return
    _all_reduce(grad_norm_accum, dist.ReduceOp.SUM)
    _all_reduce(vis_counts, dist.ReduceOp.SUM)
    _all_reduce(max_2d_size, dist.ReduceOp.MAX)


def broadcast_randn(shape, device) -> Tensor:
    """``torch.randn(shape)`` drawn on rank 0 and broadcast: the noise of ``split_gaussians`` (models/gaussians/vanilla.py:343) has to be
    the same on every replica, or the replicas' Gaussians diverge at the first densification ("extra state")."""
    t = torch.randn(tuple(shape), device=device) if (not _active() or dist.get_rank() == 0) else torch.empty(tuple(shape), device=device)
    if _active():
        ISSUED["broadcast"] += 1
        dist.broadcast(t, src=0)
    return t


def _active() -> bool:
    """A process group is up and its collectives are to be issued: several ranks, or ONE rank with ``force_collectives``."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or collectives_forced())


def refinement_after_synced(model, step: int, optimizer, stats=None, verbose: bool = False) -> None:
    """View-parallel densification: sum / max the per-view statistics over the ranks (``reduce_densify_stats``), then let every rank
    run the same refinement on its replica with rank 0's noise.  Parameters, optimiser state and reduced statistics are identical on
    all ranks, hence so are the plan and the resulting set.  ``stats``: an ``optim.DensifyStats`` (or None if ``model`` already
    carries ``xys_grad_norm / vis_counts / max_2Dsize``).  Create the per-rank statistics with
    ``DensifyStats(..., first_call_initialises=(rank == 0))``: the reference's first ``after_train`` call sets ``vis_counts`` to one
    for EVERY Gaussian; only one rank may contribute that one, then the sums equal a single process visiting the same views."""
    from .densify import refinement_after
    if stats is not None:
        model.xys_grad_norm, model.vis_counts, model.max_2Dsize = stats.xys_grad_norm, stats.vis_counts, stats.max_2Dsize
    if model.xys_grad_norm is not None:
        reduce_densify_stats(model.xys_grad_norm, model.vis_counts, model.max_2Dsize)
    refinement_after(model, step, optimizer, verbose=verbose, sample_fn=broadcast_randn)


def view_for_rank(step: int, rank: int, world: int, n_views: int) -> int:
    """Round-robin view assignment: at step s, rank r renders view (s*world + r) mod n_views."""
    return (step * world + rank) % n_views


def measure_busbw(device, nbytes: int = 256 << 20, iters: int = 3) -> float:
    """Bus bandwidth (bytes / s, the rccl-tests convention: 2 (n-1)/n x message / time) of a SUM all-reduce of ``nbytes`` over the
    process group as it is -- RCCL over xGMI on a multi-GPU node, gloo when the ranks share a device.  Collective; 0.0 at world size 1."""
    if not _active():
        return 0.0
    import time
    n = dist.get_world_size()
    buf = torch.zeros(nbytes // 4, device=device, dtype=torch.float32)
    _all_reduce(buf, dist.ReduceOp.SUM)                      # (connection set-up)
    if buf.is_cuda:
        torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        _all_reduce(buf, dist.ReduceOp.SUM)
    if buf.is_cuda:
        torch.cuda.synchronize(device)
    dt = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64, device=device)
    _all_reduce(dt, dist.ReduceOp.MAX)
    return 2.0 * (n - 1) / n * nbytes / float(dt[0])


def dynamic_union_bound(n_vis_max: int, n_rows: int, device=None) -> int:
    """Exchange capacity of a REPLAYABLE view slot (graph_view.FrameGraph(dynamic=True, exchange=..)): every rank draws its own random
    camera per step, so the union of a slot's visible sets over the ranks is bounded by the SUM of the ranks' largest visible sets
    (each rank's maximum over its calibration sweep), capped at the scene.  A collective (SUM all-reduce of one integer) when a
    process group is up: every rank gets the same number.  ``device``: where the one-element tensor lives (None: CPU / gloo)."""
    total = int(n_vis_max)
    if _active():
        t = torch.tensor([total], dtype=torch.int64, device=device if device is not None else "cpu")
        _all_reduce(t, dist.ReduceOp.SUM)
        total = int(t[0])
    return min(total, int(n_rows))


def union_row_counts(masks: Iterable[Tensor]) -> List[int]:
    """Per view: how many Gaussians at least one rank sees (``masks``: this rank's uint8 / bool visibility mask of every view of its
    frame, [N] each).  Collective; identical on every rank."""
    out = []
    for m in masks:
        u = m.reshape(-1).to(torch.uint8).clone()
        if _active():
            _all_reduce(u, dist.ReduceOp.MAX)
        out.append(int(u.sum()))
    return out


def plan_exchange(union_counts: Iterable[int], n_rows: int, row_floats: int, tail_floats: int, world: int, frame_seconds: float,
                  busbw: float, headroom: float = 1.25) -> dict:
    """Price the two ways a frame's gradients can be summed over ``world`` ranks, from measured inputs only: the per-view unions of
    the ranks' visible sets (``union_row_counts``), one rank's compute time per frame and the fabric's all-reduce bus bandwidth
    (``measure_busbw``).

      per view  -- V all-reduces of ``headroom x max union`` compact rows each; all but the last run behind the following view's
                   compute, so the exposed time is what the wire needs beyond (V-1)/V of the frame + the last view's exchange;
      per frame -- one all-reduce of the dense buffer, fully exposed.

    Returns both prices and ``per_view`` = the cheaper one (``FrameExchange(per_view=...)``).  Deterministic in its inputs, which are
    identical on every rank when ``frame_seconds`` / ``busbw`` were max-reduced over the ranks."""
    counts = [int(c) for c in union_counts]
    V = max(len(counts), 1)
    ring = 2.0 * (world - 1) / max(world, 1)
    cap = min(int(max(counts, default=0) * headroom) + 64, n_rows)
    view_bytes = cap * row_floats * 4
    dense_bytes = (n_rows * row_floats + tail_floats) * 4
    bw = max(float(busbw), 1.0)
    wire_view = ring * view_bytes / bw                     # seconds per view's exchange
    hidden = frame_seconds * (V - 1) / V
    exposed_view = max(0.0, (V - 1) * wire_view - hidden) + wire_view + ring * tail_floats * 4 / bw
    exposed_frame = ring * dense_bytes / bw
    return {"per_view": bool(exposed_view <= exposed_frame), "exposed_per_view_ms": exposed_view * 1e3, "exposed_per_frame_ms": exposed_frame * 1e3,
            "per_view_bytes": V * view_bytes + tail_floats * 4, "per_frame_bytes": dense_bytes, "busbw_GBps": bw / 1e9,
            "frame_compute_ms": frame_seconds * 1e3, "union_fraction": max(counts, default=0) / max(n_rows, 1)}
