#!/usr/bin/env python3
"""Isolated timing of the colour transform's forward with and without the training loss on its launch (measurement tooling):
bds_bilagrid_ms_ed_fwd + bds_l1_tv_train against bds_bilagrid_ms_ed_train_fwd at 1920x1080, three levels."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from bilateral_driving_amd import _lib as L  # noqa: E402
from bilateral_driving_amd import harness as Hn  # noqa: E402
from bilateral_driving_amd.bilagrid import _levels_struct  # noqa: E402


def main():
    dev = torch.device("cuda")
    H, W = 1080, 1920
    lib, st = L.lib(), L.stream()
    grids = [g.to(dev) for g in Hn.make_grids(6, device=dev)]
    gg = [torch.zeros_like(g) for g in grids]
    sel = [g[0:1].contiguous() for g in grids]
    lv = _levels_struct(sel, None, Hn.FACTORS_3)
    tv = _levels_struct(grids, gg, [1] * len(grids))
    wts = (C.c_float * 3)(*[0.01 * 0.5 * math.sqrt(g.shape[4] * g.shape[3] * g.shape[2]) for g in grids])
    render, alphas = torch.rand(1, H, W, 4, device=dev), torch.rand(1, H, W, 1, device=dev)
    sky, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    rgb, depth, v_rgb = torch.empty(H, W, 3, device=dev), torch.empty(H, W, device=dev), torch.empty(H, W, 3, device=dev)
    loss = torch.zeros(L.LOSS_SLOTS * L.LOSS_SLOT_STRIDE, device=dev)
    nb = lib.bds_bilagrid_ms_workspace_bytes(3, lv, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def separate():
        L.check(lib.bds_bilagrid_ms_ed_fwd(3, lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(ws), nb, L.ptr(rgb), L.ptr(depth), st), "fwd")
        L.check(lib.bds_l1_tv_train(rgb.numel(), L.ptr(rgb), L.ptr(target), 3, tv, wts, 1.0, L.ptr(loss), L.LOSS_SLOTS, L.ptr(v_rgb), st), "loss")

    def fused():
        L.check(lib.bds_bilagrid_ms_ed_train_fwd(3, lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(ws), nb, L.ptr(rgb), L.ptr(depth),
                                                 L.ptr(target), 3, tv, wts, 1.0, L.ptr(loss), L.LOSS_SLOTS, L.ptr(v_rgb), st), "train_fwd")

    def plain():
        L.check(lib.bds_bilagrid_ms_ed_fwd(3, lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(ws), nb, L.ptr(rgb), L.ptr(depth), st), "fwd")

    for name, fn in (("plain transform", plain), ("transform + loss launch", separate), ("loss on the transform launch", fused)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:32s} {e0.elapsed_time(e1) / 50 * 1000:8.1f} us")


if __name__ == "__main__":
    main()
