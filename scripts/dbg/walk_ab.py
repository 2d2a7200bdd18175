"""A/B of the WALKING form of the bilateral pyramid backward (profiles/r09i_bilagrid_bwd_walk_ab.txt).  The form lost and was removed:
this script needs profiles/r09i_bilagrid_bwd_walk_form.patch applied (option 9 = rows a workgroup walks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bilateral_driving_amd import _lib as L
from bilateral_driving_amd.bilagrid import bilagrid_transform
dev = "cuda"
SHAPES = {"headline": (1080, 1920, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]), "c3": (900, 1600, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
          "c5": (1280, 1920, [(2, 2, 1), (4, 4, 2), (8, 8, 4), (16, 16, 8)], [8, 4, 4, 2]), "odd": (271, 333, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
          "f3": (540, 960, [(4, 4, 2), (8, 8, 4)], [3, 2])}
g = torch.Generator().manual_seed(0)
for name, (H, W, levels, factors) in SHAPES.items():
    rgb = torch.rand(H, W, 3, generator=g).to(dev).requires_grad_(True)
    alpha = torch.rand(H, W, generator=g).to(dev); sky = torch.rand(H, W, 3, generator=g).to(dev)
    grids = []
    for (gx, gy, gl) in levels:
        ident = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]).reshape(12, 1, 1, 1).repeat(1, gl, gy, gx)
        grids.append((ident + 0.05 * torch.randn(12, gl, gy, gx, generator=g)).to(dev).requires_grad_(True))
    wt = torch.randn(H, W, 3, generator=g).to(dev)
    def run():
        for t in [rgb] + grids: t.grad = None
        out = bilagrid_transform(rgb, grids, factors, alpha=alpha, sky=sky)
        (out * wt).sum().backward()
        return [rgb.grad.clone()] + [x.grad.clone() for x in grids]
    L.set_option(9, 0); ref = run()
    for rows in (0, 8, 16, 32, 64):
        L.set_option(9, rows)
        got = run()
        err = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(got, ref)]
        for _ in range(3): run()
        L.enable_timers(True)
        for _ in range(20): run()
        torch.cuda.synchronize(); t = L.timer_summary(); L.enable_timers(False)
        print(f"{name:9s} walk={rows:3d} max rel err vs walk 0: {max(err):.1e}  " + "  ".join(f"{k} {v[1] * 1e3:7.1f} us" for k, v in sorted(t.items())), flush=True)
L.set_option(9, 0)
