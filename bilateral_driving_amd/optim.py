"""Drop-in for the optimiser of the reference trainer: ``torch.optim.Adam(groups, lr=0.0, eps=1e-15)``
(/root/reference/project/models/trainers/base.py:222) with the parameter update done by ONE HIP pass per tensor
(``bds_adam_step``) instead of torch's multi-kernel foreach implementation.  Same constructor arguments, same
``param_groups`` (the reference's LR schedulers write ``group["lr"]``, base.py:240-246) and the same ``state`` layout
(``step``, ``exp_avg``, ``exp_avg_sq``) that the reference's densification code edits in place
(models/gaussians/basics.py:162-206), so it can replace the optimiser without touching the trainer."""
from __future__ import annotations

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    MULTI_MAX_TENSORS, MULTI_MAX_ELEMS = 12, 1 << 24     # bds_adam_step_multi: tensors per launch; larger tensors take their own pass

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 consume_grads: bool = False):
        """``consume_grads``: every gradient is cleared by the pass that reads it (bds_adam_step_consume) -- for loops whose backward
        accumulates into persistent ``.grad`` buffers (``graph_view.FrameGraph(clear_grads=False)``): no ``zero_grad()`` pass."""
        self.consume_grads = bool(consume_grads)
        # the small groups (xyz, rotation, scaling, opacity, the grids of every level, ...) step in ONE launch (include/bds.h
        # bds_adam_step_multi: same arithmetic per element); BDS_ADAM_MULTI=0: one launch per tensor
        self.multi_tensor = __import__("os").environ.get("BDS_ADAM_MULTI", "1") == "1"
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _step_rowblock(self, base: int, n_rows: int, members) -> None:
        import ctypes as C
        n = len(members)
        vp = lambda vals: (C.c_void_p * n)(*vals)
        dbl = lambda i: (C.c_double * n)(*[b[i] for b in members])
        L.check(L.lib().bds_adam_step_rowblock(
            n_rows, base, n, vp([b[0].data_ptr() for b in members]), vp([b[2].data_ptr() for b in members]), vp([b[3].data_ptr() for b in members]),
            (C.c_int * n)(*[(b[1] - base) // 4 for b in members]), (C.c_int * n)(*[b[5] for b in members]), dbl(7), dbl(8), dbl(9), dbl(10), dbl(11),
            (C.c_int64 * n)(*[b[12] for b in members]), int(self.consume_grads), L.stream()), "bds_adam_step_rowblock")

    def _step_multi(self, batch) -> None:
        import ctypes as C
        n = len(batch)
        if n == 0:
            return
        vp = lambda vals: (C.c_void_p * n)(*vals)
        dbl = lambda i: (C.c_double * n)(*[b[i] for b in batch])
        L.check(L.lib().bds_adam_step_multi(
            n, vp([b[0].data_ptr() for b in batch]), vp([b[1] for b in batch]), vp([b[2].data_ptr() for b in batch]),
            vp([b[3].data_ptr() for b in batch]), (C.c_int64 * n)(*[b[4] for b in batch]), (C.c_int * n)(*[b[5] for b in batch]),
            (C.c_int64 * n)(*[b[6] for b in batch]), dbl(7), dbl(8), dbl(9), dbl(10), dbl(11), (C.c_int64 * n)(*[b[12] for b in batch]),
            int(self.consume_grads), L.stream()), "bds_adam_step_multi")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, st = L.lib(), L.stream()
        batch = []      # (p, grad address, m, v, n, width, grad stride, lr, b1, b2, eps, wd, step): tensors stepped by ONE launch
        for group in self.param_groups:
            b1, b2 = group["betas"]
            if group.get("deferred_rows"):       # (DeferredRowAdam steps these through the views' visible-id lists)
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                L.require_gpu(p)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam expects contiguous float32 parameters")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0)   # host scalar, as torch's default (capturable=False) layout
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.shape == p.shape and v.shape == p.shape):
                    raise RuntimeError("optimizer state does not match its parameter (after densification, re-create both)")
                gr = p.grad
                hyper = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), int(state["step"]))
                if (not gr.is_contiguous() and gr.dim() in (1, 2) and gr.shape == p.shape and (gr.dim() == 1 or gr.stride(1) == 1)
                        and gr.stride(0) >= (1 if gr.dim() == 1 else gr.shape[1])):
                    # a column range of a row block (dist.FlatGradients(row_block=True)): read -- and cleared -- where it lies
                    width = 1 if gr.dim() == 1 else int(gr.shape[1])
                    if self.multi_tensor:
                        batch.append((p, gr.data_ptr(), m, v, p.numel(), width, int(gr.stride(0)), *hyper))
                        continue
                    L.check(lib.bds_adam_step_rows(p.shape[0], width, int(gr.stride(0)), L.ptr(p), gr.data_ptr(), L.ptr(m), L.ptr(v),
                                                   *hyper[:5], hyper[5], int(self.consume_grads), st), "bds_adam_step_rows")
                    continue
                g = p.grad.contiguous()
                consume = self.consume_grads and g.data_ptr() == p.grad.data_ptr()     # (clearing a contiguous COPY would clear nothing)
                if self.multi_tensor and p.numel() <= self.MULTI_MAX_ELEMS and (consume or not self.consume_grads):
                    batch.append((p, g.data_ptr(), m, v, p.numel(), 0, 0, *hyper))      # (large tensors keep the 16-byte-vector pass)
                    if g is not p.grad:
                        batch[-1] = batch[-1] + (g,)     # (keep the contiguous copy alive until the launch)
                    continue
                fn = lib.bds_adam_step_consume if consume else lib.bds_adam_step
                L.check(fn(p.numel(), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), *hyper, st), "bds_adam_step")
                if self.consume_grads and not consume:
                    p.grad.zero_()
        # tensors whose gradients are column ranges of ONE [N,16] row block (dist.FlatGradients(row_block=True)) step together through
        # the block, which is then read -- and cleared -- once (bds_adam_step_rowblock) instead of once per tensor
        blocks = {}
        for b in batch:
            if b[5] > 0 and b[5] <= 4 and b[6] == 16 and b[0].dim() >= 1:
                blocks.setdefault((b[1] & ~63, b[0].shape[0]), []).append(b)
        for (base, n_rows), members in blocks.items():
            if 2 <= len(members) <= 4 and base % 16 == 0:
                self._step_rowblock(base, n_rows, members)
                batch = [b for b in batch if not any(b is m for m in members)]
        for k in range(0, len(batch), self.MULTI_MAX_TENSORS):
            self._step_multi(batch[k:k + self.MULTI_MAX_TENSORS])
        return loss


class DeferredRowAdam(FusedAdam):
    """``FusedAdam`` whose groups marked ``deferred_rows=True`` are stepped ROW-LAZILY with the dense pass's numbers, bit for bit.

    The reference steps one dense ``torch.optim.Adam`` after every single-view iteration
    (/root/reference/project/tools/train.py:252-283 -> models/trainers/base.py:502-516, optimizer at :222-226): 28 bytes per
    parameter float per step, 81 % of them SH coefficients of Gaussians the view never saw (their gradient rows are exact zeros, but a
    zero-gradient row still moves: m <- b1 m, v <- b2 v, p <- p - lr_t m^/(sqrt(v^)+eps)).  A deferred parameter keeps, per row, the
    step it is current for; the steps it missed are replayed -- same fp32 operations, same order, per-step lr and bias corrections from
    a small device table -- when the row is next read (``catchup``: enqueued by the view's forward between the visible-id list and the
    record pack, ``graph_view.FrameGraph(row_catchup=opt.catchup)``) and stepped from its gradient row by ``step(lists=...)`` over the
    frame's visible-id lists (``FrameGraph.row_lists()``).  Adam's bytes per step are then proportional to the visible rows.

    Only parameters that nothing but a view's list-driven kernels read may be deferred: the SH coefficients ([N,K,3], or the reference's
    ``_features_dc`` [N,3] / ``_features_rest`` [N,K-1,3]).  Means / quats / scales decide visibility and the opacity is activated by
    the projection pass over ALL rows: they stay dense.  ``flush()`` brings every row up to date -- call it before anything else reads
    a deferred tensor (densification: ``before_refinement()``; checkpoints: ``state_dict()`` does; evaluation through another path);
    it also runs by itself every ``table_steps - 2`` steps (the table is a ring).

    Group keys: ``deferred_rows`` (bool); ``col_split`` / ``lr_b``: columns >= col_split of a row (the flattened trailing dimensions)
    use ``lr_b`` instead of ``lr`` (one [N,16,3] tensor holding dc + rest: col_split = 3, lr_b = the rest's rate)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 consume_grads: bool = False, table_steps: int = 1024, dense_above: float = 0.6):
        """``dense_above``: a step whose lists' capacities add up to at least this fraction of the rows runs as ONE pass over all rows."""
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, consume_grads=consume_grads)
        self.dense_above = float(dense_above)
        assert table_steps >= 8
        self.table_steps = int(table_steps)
        self._t = 0                      # steps taken by the deferred parameters
        self._last_flush = 0
        self._clock = None               # int32 [1] on the device: the step the deferred parameters are at (read by captured catch-ups)
        for group in self.param_groups:
            if group.get("deferred_rows"):
                for p in group["params"]:
                    self._ensure(p, group)

    # ---- state ---------------------------------------------------------------------------------------------------------------------
    def _deferred(self):
        for group in self.param_groups:
            if group.get("deferred_rows"):
                for p in group["params"]:
                    yield group, p

    def _ensure(self, p, group):
        L.require_gpu(p)
        if p.dtype != torch.float32 or not p.is_contiguous() or p.dim() < 1:
            raise RuntimeError("DeferredRowAdam expects contiguous float32 parameters [N, ...]")
        N = p.shape[0]
        rf = p.numel() // max(N, 1)
        if not 1 <= rf <= 256:
            raise RuntimeError(f"deferred rows of {rf} floats (1 .. 256 supported)")
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.tensor(float(self._t))
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if "last_step" not in st or st["last_step"].shape[0] != N:
            # (after densification surgery: every row is current -- ``before_refinement`` flushed -- and so are the new ones)
            st["last_step"] = torch.full((N,), self._t, device=p.device, dtype=torch.int32)
        if "table" not in st:
            st["table"] = torch.zeros(self.table_steps, 4, device=p.device, dtype=torch.float32)
            st["hyper"] = (tuple(group["betas"]), float(group["eps"]), float(group["weight_decay"]))
        if self._clock is None:
            self._clock = torch.full((1,), self._t, device=p.device, dtype=torch.int32)
        if not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous() and st["exp_avg"].shape == p.shape):
            raise RuntimeError("optimizer state does not match its parameter (after densification, re-create both)")
        return st, N, rf

    def _advance(self, group, p, n_cap, n_dev, ids, with_step: bool, clock: bool):
        st, N, rf = self._ensure(p, group)
        (b1, b2), eps, wd = st["hyper"]
        split = int(group.get("col_split", rf))
        g = None
        if with_step:
            g = p.grad
            if g is None or not g.is_contiguous() or g.shape != p.shape:
                raise RuntimeError("a deferred parameter needs a contiguous dense .grad buffer (dist.FlatGradients)")
        L.check(L.lib().bds_adam_rows_advance(int(n_cap), n_dev, ids, N, rf, split, L.ptr(p), None if g is None else g.data_ptr(),
                                              int(self.consume_grads), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.ptr(st["last_step"]),
                                              L.ptr(self._clock) if (clock or with_step) else None, self._t, int(with_step), L.ptr(st["table"]),
                                              self.table_steps, float(group["lr"]), float(group.get("lr_b", group["lr"])), float(b1), float(b2),
                                              eps, wd, L.stream()), "bds_adam_rows_advance")

    # ---- the three uses ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def catchup(self, n_cap: int, n_dev, ids) -> None:
        """Bring the rows of a view's visible-id list (``ids``: device pointer to int32 [n_cap]; ``n_dev``: device pointer to the
        count, or None) to the optimizer's current step -- read from the device-side clock, so a captured launch stays valid."""
        ids = ids.data_ptr() if torch.is_tensor(ids) else ids
        for group, p in self._deferred():
            self._advance(group, p, n_cap, n_dev, ids, with_step=False, clock=True)

    @torch.no_grad()
    def flush(self) -> None:
        """Every row of every deferred parameter up to the current step (dense pass; replays only what is missing)."""
        for group, p in self._deferred():
            self._advance(group, p, p.shape[0], None, None, with_step=False, clock=False)
        self._last_flush = self._t

    def before_refinement(self) -> None:
        """Densification reads and re-allocates the parameters and their moments: make them current first."""
        self.flush()

    def state_dict(self):
        self.flush()
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        """A checkpoint written by ``state_dict()`` holds every row at the step of the save (it flushes): the rows' step words are made
        again (torch's loader casts every state tensor to the parameter's dtype), the clock and the step count follow ``state["step"]``."""
        super().load_state_dict(state_dict)
        for group, p in self._deferred():
            st = self.state[p]
            self._t = int(st["step"])
            st["last_step"] = torch.full((p.shape[0],), self._t, device=p.device, dtype=torch.int32)
            st["table"] = st["table"].to(device=p.device, dtype=torch.float32).contiguous() if torch.is_tensor(st.get("table")) else \
                torch.zeros(self.table_steps, 4, device=p.device, dtype=torch.float32)
            if st["table"].shape != (self.table_steps, 4):
                st["table"] = torch.zeros(self.table_steps, 4, device=p.device, dtype=torch.float32)     # (every row is current: no entry is needed)
            st["hyper"] = (tuple(group["betas"]), float(group["eps"]), float(group["weight_decay"]))
            if self._clock is None:
                self._clock = torch.full((1,), self._t, device=p.device, dtype=torch.int32)
            self._clock.fill_(self._t)
        self._last_flush = self._t

    @torch.no_grad()
    def step(self, closure=None, lists=None):
        """``lists``: [(n_capacity, n_dev_ptr | None, ids)] -- the visible-id lists of the views whose gradients are in ``.grad``
        (``FrameGraph.row_lists()``); a row listed twice steps once.  None: every row takes the step (the dense pass, same numbers)."""
        loss = super().step(closure)          # the dense groups
        deferred = list(self._deferred())
        if not deferred:
            return loss
        for group, p in deferred:             # hyper-parameters other than lr are baked into the replay: a change flushes first
            st, _, _ = self._ensure(p, group)
            now = (tuple(group["betas"]), float(group["eps"]), float(group["weight_decay"]))
            if st["hyper"] != now:
                self.flush()
                st["hyper"] = now
        self._t += 1
        dense_step = False
        for group, p in deferred:
            self.state[p]["step"] += 1
            if lists is None or sum(int(l[0]) for l in lists) >= self.dense_above * p.shape[0]:
                # every row takes the step (rows that are behind replay first: the same numbers) -- also when the lists' capacities
                # add up to most of the tensor (a frame of several views): one pass over whole rows beats one pass per list
                self._advance(group, p, p.shape[0], None, None, with_step=True, clock=True)
                dense_step = True
            else:
                for n_cap, n_dev, ids in lists:
                    self._advance(group, p, n_cap, n_dev, ids.data_ptr() if torch.is_tensor(ids) else ids, with_step=True, clock=True)
                if not lists:                 # (no view this step: the step's table entry and the clock still have to be written)
                    self._advance(group, p, 0, None, None, with_step=True, clock=True)
        if dense_step:
            self._last_flush = self._t
        elif self._t - self._last_flush >= self.table_steps - 2:
            self.flush()
        return loss


class DensifyStats:
    """Accumulators of the reference's adaptive density control (``VanillaGaussians.xys_grad_norm / vis_counts / max_2Dsize``,
    models/gaussians/vanilla.py:163-191), updated by one launch per step instead of ~10 masked-indexing operations with host
    syncs.  ``update(info)`` takes the ``info`` dict of ``rasterization()`` / ``fused_view`` after ``backward()``."""

    def __init__(self, num_points: int, device, batch_size: int = 1, first_call_initialises: bool = True):
        """``first_call_initialises=False`` (view-parallel training, every rank but one): start from zeros and count only visible
        Gaussians from the first call on, so that the sum over the ranks equals one process visiting all the views
        (``dist.refinement_after_synced``)."""
        self.xys_grad_norm = torch.zeros(num_points, device=device, dtype=torch.float32)
        self.vis_counts = torch.zeros(num_points, device=device, dtype=torch.float32)
        self.max_2Dsize = torch.zeros(num_points, device=device, dtype=torch.float32)
        self.batch_size = int(batch_size)
        self._first = bool(first_call_initialises)

    @torch.no_grad()
    def update(self, info, absgrad: bool = True) -> None:
        m2 = info["means2d"]
        g = m2.absgrad if absgrad else m2.grad
        radii = info["radii"]
        assert g is not None, "call after backward() (and with absgrad=True / retain_grad() as the trainer does)"
        g, radii = g.reshape(-1, 2).contiguous(), radii.reshape(-1).contiguous()
        N = radii.numel()
        assert N == self.xys_grad_norm.numel(), "one camera, all Gaussians (the reference uses batch_size 1)"
        W, H = int(info["width"]), int(info["height"])
        L.check(L.lib().bds_densify_stats(N, L.ptr(g), L.ptr(radii), W, H, self.batch_size, max(W, H), int(self._first),
                                          L.ptr(self.xys_grad_norm), L.ptr(self.vis_counts), L.ptr(self.max_2Dsize), L.stream()),
                "bds_densify_stats")
        self._first = False
