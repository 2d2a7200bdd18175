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
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 consume_grads: bool = False):
        """``consume_grads``: every gradient is cleared by the pass that reads it (bds_adam_step_consume) -- for loops whose backward
        accumulates into persistent ``.grad`` buffers (``graph_view.FrameGraph(clear_grads=False)``): no ``zero_grad()`` pass."""
        self.consume_grads = bool(consume_grads)
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, st = L.lib(), L.stream()
        for group in self.param_groups:
            b1, b2 = group["betas"]
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
                if (not gr.is_contiguous() and gr.dim() in (1, 2) and gr.shape == p.shape and (gr.dim() == 1 or gr.stride(1) == 1)
                        and gr.stride(0) >= (1 if gr.dim() == 1 else gr.shape[1])):
                    # a column range of a row block (dist.FlatGradients(row_block=True)): read -- and cleared -- where it lies
                    width = 1 if gr.dim() == 1 else int(gr.shape[1])
                    L.check(lib.bds_adam_step_rows(p.shape[0], width, int(gr.stride(0)), L.ptr(p), gr.data_ptr(), L.ptr(m), L.ptr(v),
                                                   float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                   int(state["step"]), int(self.consume_grads), st), "bds_adam_step_rows")
                    continue
                g = p.grad.contiguous()
                consume = self.consume_grads and g.data_ptr() == p.grad.data_ptr()     # (clearing a contiguous COPY would clear nothing)
                fn = lib.bds_adam_step_consume if consume else lib.bds_adam_step
                L.check(fn(p.numel(), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), float(group["lr"]), float(b1), float(b2),
                           float(group["eps"]), float(group["weight_decay"]), int(state["step"]), st), "bds_adam_step")
                if self.consume_grads and not consume:
                    p.grad.zero_()
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
