#!/usr/bin/env python3
"""Where the column-strip backward of the bilateral transform spends its time: the kernel alone at the benchmark's size (1080p,
3 levels, noisy guidance as in bench.py) with parts switched off through the ablation mask (bds_set_option(3, mask)), HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
from bilateral_driving_amd.bilagrid import _levels_struct

lib = L.lib()
H, W = 1080, 1920
g = torch.Generator().manual_seed(5)
render = (torch.rand(H, W, 4, generator=g) * 1.1).cuda()
alpha = torch.rand(H, W, generator=g).cuda()
sky = torch.rand(H, W, 3, generator=g).cuda()
smooth = "--smooth" in sys.argv
if smooth:   # a smooth image: neighbouring cells share their guidance bin
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    render[..., 0], render[..., 1], render[..., 2] = xx.cuda(), yy.cuda(), (0.5 * xx + 0.5 * yy).cuda()
    alpha.fill_(1.0)
grids = [x[1:2].contiguous().cuda() for x in Hn.make_grids(3, seed=4)]
factors = Hn.FACTORS_3
lv = _levels_struct(grids, None, factors)
wsb = lib.bds_bilagrid_ms_workspace_bytes(3, lv, H, W)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
rgb, depth = torch.empty(H, W, 3, device="cuda"), torch.empty(H, W, device="cuda")
st = L.stream()
v_out = torch.randn(H, W, 3, generator=g).cuda()
vg = [torch.zeros_like(x) for x in grids]
lvb = _levels_struct(grids, vg, factors)
v_render, v_alpha, v_sky = torch.empty(H, W, 4, device="cuda"), torch.empty(H, W, device="cuda"), torch.empty(H, W, 3, device="cuda")


def run(mask, rows=0, reps=20):
    L.set_option(3, mask); L.set_option(5, rows); L.set_option(7, 3)
    def fwd():
        L.check(lib.bds_bilagrid_ms_ed_fwd(3, lv, H, W, L.ptr(render), L.ptr(alpha), L.ptr(sky), L.ptr(ws), wsb, L.ptr(rgb), L.ptr(depth), st), "fwd")
    def bwd():
        L.check(lib.bds_bilagrid_ms_ed_bwd(3, lvb, H, W, L.ptr(render), L.ptr(alpha), L.ptr(sky), L.ptr(ws), wsb, L.ptr(v_out), None, None,
                                           L.ptr(v_render), L.ptr(v_alpha), L.ptr(v_sky), st), "bwd")
    out = []
    for fn in (fwd, bwd):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps * 1e3)
    L.set_option(3, 0); L.set_option(5, 0)
    return out

print("image:", "smooth" if smooth else "noise")
for name, mask in (("general kernels", 16), ("strips", 0), ("strips, no scatter", 32), ("strips, no guidance", 64), ("strips, no scatter/guidance", 96),
                   ("strips, no x-reduce/scatter/guidance", 224), ("strips, no completed-row work", 256)):
    f, b = run(mask)
    print(f"{name:40s} fwd {f:7.1f} us   bwd {b:7.1f} us")
for rows in (16, 24, 32, 48, 64):
    f, b = run(0, rows)
    print(f"strips, rows per band {rows:3d}               fwd {f:7.1f} us   bwd {b:7.1f} us")
