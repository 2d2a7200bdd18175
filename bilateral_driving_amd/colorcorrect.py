"""``color_correct(img, ref, num_iters=5, eps=0.5/255)`` of the reference's bilateral/lib_bilagrid.py:56-120 (the evaluation
path applies it to every rendered frame, models/video_utils_color_correction.py:201) on MI355X: per iteration ONE streaming HIP
pass (apply the previous quadratic warp, clip, accumulate the three masked 10x10 normal equations in double) and a float64 solve of
the three small systems on the device -- no host sync, no [pixels x 10] matrix, no QR of it."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib as L

_IU = torch.triu_indices(10, 10)


@torch.no_grad()
def color_correct(img: Tensor, ref: Tensor, num_iters: int = 5, eps: float = 0.5 / 255) -> Tensor:
    if img.shape[-1] != ref.shape[-1]:
        raise ValueError(f"img's {img.shape[-1]} and ref's {ref.shape[-1]} channels must match")
    if img.shape[-1] != 3:
        raise NotImplementedError("color_correct is built for 3 channels (RGB)")
    L.require_gpu(img, ref)
    dev = img.device
    cur = img.detach().reshape(-1, 3).contiguous().float()
    rf = ref.detach().reshape(-1, 3).contiguous().float()
    if rf.shape != cur.shape:
        raise ValueError("img and ref must have the same number of pixels")
    P = cur.shape[0]
    mask0 = torch.empty(P, dtype=torch.uint8, device=dev)
    bufs = [torch.empty_like(cur), torch.empty_like(cur)] if num_iters > 0 else []      # the input is never written
    acc = torch.zeros(3, 65, dtype=torch.float64, device=dev)
    iu = _IU.to(dev)
    lib, st = L.lib(), L.stream()
    warp = None
    for it in range(num_iters + 1):
        last = it == num_iters
        if not last:
            acc.zero_()
        # pass `it`: cur_it = it ? clip(expand(cur_{it-1}) @ warp_{it-1}) : img; statistics of cur_it for warp_it
        nxt = bufs[it % 2] if it else None
        if last and not it:
            break                                                         # num_iters == 0: the input itself
        L.check(lib.bds_color_correct_step(P, L.ptr(cur), L.ptr(rf), L.ptr(warp), float(eps), L.ptr(mask0), L.ptr(nxt),
                                           None if last else L.ptr(acc), st), "bds_color_correct_step")
        if it:
            cur = nxt
        if last:
            break
        G = torch.zeros(3, 10, 10, dtype=torch.float64, device=dev)
        G[:, iu[0], iu[1]] = acc[:, :55]
        G = G + G.transpose(1, 2) - torch.diag_embed(torch.diagonal(G, dim1=1, dim2=2))
        h = acc[:, 55:, None]
        # least squares solution of the (possibly rank-deficient) normal equations; the reference calls lstsq on the tall system
        w = torch.linalg.pinv(G, hermitian=True) @ h
        if not bool(torch.isfinite(w).all()):
            raise AssertionError("color_correct: non-finite warp")      # the reference asserts the same (:113)
        warp = w[:, :, 0].transpose(0, 1).contiguous().float()          # [10, 3]
    return cur.reshape(img.shape).clone() if num_iters == 0 else cur.reshape(img.shape)
