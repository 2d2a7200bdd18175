"""Multi-GPU data parallelism over VIEWS (new capability: the reference trains one view per step on
one GPU -- /root/reference/project/models/trainers/base.py:411, scripts/train.sh:2,23).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  Every rank holds a full replica of the Gaussian parameters and renders its own view; the only
exchange step of an iteration is ONE sum-all-reduce of the flat per-Gaussian gradient buffer
(59 floats/Gaussian at SH degree 3: 3 means + 4 quats + 3 scales + 1 opacity + 48 SH) plus the small
bilateral-grid / pose tail.  Parameter gradients live as views into that flat buffer, so there is no
pack/unpack pass: autograd accumulates straight into the communication buffer.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class FlatGradients:
    """Owns one contiguous fp32 buffer; ``param.grad`` of every registered parameter is a view into it."""

    def __init__(self, params: Iterable[Tensor]):
        self.params: List[Tensor] = [p for p in params]
        assert self.params, "no parameters"
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        self.views: List[Tensor] = []
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev and p.requires_grad
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += p.numel()
        self._work = None

    def zero(self) -> None:
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:  # someone replaced .grad (e.g. optimizer.zero_grad(set_to_none=True))
                p.grad = v

    def check_views(self) -> None:
        for p, v in zip(self.params, self.views):
            assert p.grad is not None and p.grad.data_ptr() == v.data_ptr(), "param.grad no longer aliases the flat buffer"

    def all_reduce(self, average: bool = False, async_op: bool = False):
        """Sum (or average) the gradients over all ranks.  No-op for world size 1."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        self.check_views()
        self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op and average:
            self.flat.div_(dist.get_world_size())
        return self._work

    def wait(self, average: bool = False) -> None:
        if self._work is not None:
            self._work.wait()
            self._work = None
            if average:
                self.flat.div_(dist.get_world_size())

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


def reduce_densify_stats(grad_norm_accum: Tensor, vis_counts: Tensor, max_2d_size: Tensor) -> None:
    """Per-view densification statistics (models/gaussians/vanilla.py:163-191) become global:
    sums for the accumulated gradient norm / visibility counts, max for the screen-space size."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(vis_counts, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_2d_size, op=dist.ReduceOp.MAX)


def view_for_rank(step: int, rank: int, world: int, n_views: int) -> int:
    """Round-robin view assignment: at step s, rank r renders view (s*world + r) mod n_views."""
    return (step * world + rank) % n_views
