#!/usr/bin/env python3
"""Where the colour transform's backward spends its time (measurement tooling): bds_bilagrid_ms_ed_bwd at 1920x1080, three levels
(grids 2x2x1 / 4x4x2 / 8x8x4, factors 4 / 4 / 2), on a smooth synthetic render, with the ablation bits of option 3
(1: no y pass, 2: no grid scatter, 4: no guidance route)."""
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
    sel = [g[0:1].contiguous() for g in grids]
    gg = [torch.zeros_like(g) for g in sel]
    lv = _levels_struct(sel, gg, Hn.FACTORS_3)
    ys, xs = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    base = torch.stack([0.5 + 0.4 * torch.sin(7 * xs + 3 * ys), 0.5 + 0.4 * torch.cos(5 * ys - 2 * xs), 0.5 + 0.3 * torch.sin(11 * xs * ys),
                        2.0 + xs], dim=-1)
    render = (base + 0.02 * torch.randn(H, W, 4, device=dev)).reshape(1, H, W, 4).contiguous()
    alphas = torch.full((1, H, W, 1), 0.9, device=dev)
    sky = torch.rand(H, W, 3, device=dev)
    v_rgb = torch.randn(H, W, 3, device=dev) / (3 * H * W)
    v_render, v_alpha, v_sky = torch.empty(1, H, W, 4, device=dev), torch.empty(1, H, W, 1, device=dev), torch.empty(H, W, 3, device=dev)
    rgb, depth = torch.empty(H, W, 3, device=dev), torch.empty(H, W, device=dev)
    nb = lib.bds_bilagrid_ms_workspace_bytes(3, lv, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    L.check(lib.bds_bilagrid_ms_ed_fwd(3, lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(ws), nb, L.ptr(rgb), L.ptr(depth), st), "fwd")

    def bwd():
        L.check(lib.bds_bilagrid_ms_ed_bwd(3, lv, H, W, L.ptr(render), L.ptr(alphas), L.ptr(sky), L.ptr(ws), nb, L.ptr(v_rgb), None, None,
                                           L.ptr(v_render), L.ptr(v_alpha), L.ptr(v_sky), st), "bwd")

    for name, mask in (("full", 0), ("full, general guidance gather", 2048), ("full, general x / y pass taps", 4096), ("no y pass", 1), ("no grid scatter", 2), ("no guidance route", 4), ("none of the three", 7)):
        lib.bds_set_option(L.OPT_DEBUG, mask)
        for _ in range(5):
            bwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:32s} {e0.elapsed_time(e1) / 40 * 1000:8.1f} us")
    lib.bds_set_option(L.OPT_DEBUG, 0)


if __name__ == "__main__":
    main()
