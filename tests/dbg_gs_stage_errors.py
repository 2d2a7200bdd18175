"""Where a gradient error of the random sweep (tests/test_gpu_25_gs_random_sweep.py) comes from: the projection backward and the
compositor backward each ALONE, on identical inputs and incoming gradients, HIP (fp32) and the oracle in float32 both against the
oracle in float64.  Not collected by pytest; run on the GPU box:

    python tests/dbg_gs_stage_errors.py 16 20 35 43"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gs_oracle as G                                   # noqa: E402
from tests.test_gpu_25_gs_random_sweep import random_scene          # noqa: E402
from tests.util import grad_errors                                  # noqa: E402

import bilateral_driving_amd.gs_ops as ops                           # noqa: E402

NAMES = ("means", "quats", "scales")


def fmt(t):
    return "(" + ", ".join(f"{x:.1e}" for x in t) + ")"


def main(seed):
    sc, W, H, mode, kw, bg = random_scene(seed)
    N = sc["means"].shape[0]
    print(f"seed {seed}: N {N} {W}x{H} {mode} {kw}")
    vm, K = sc["viewmats"], sc["Ks"]

    def oracle_project(dt):
        inp = {k: sc[k].detach().clone().to(dt).requires_grad_(True) for k in NAMES}
        radii, m2, dep, con, _ = G.project(inp["means"], inp["quats"], inp["scales"], vm[0].to(dt), K[0].to(dt), W, H, 0.3, kw["near_plane"], 1e10,
                                           kw["radius_clip"])
        return inp, radii, m2, dep, con

    # ---- HIP projection forward: the integer decisions ----
    gin = {k: sc[k].cuda().requires_grad_(True) for k in NAMES}
    radii_g, m2_g, dep_g, con_g, _ = ops.fully_fused_projection(gin["means"], gin["quats"], gin["scales"], vm.cuda(), K.cuda(), W, H, 0.3,
                                                                kw["near_plane"], 1e10, kw["radius_clip"])
    in64, radii64, m2_64, dep64, con64 = oracle_project(torch.float64)
    in32, _, m2_32, dep32, con32 = oracle_project(torch.float32)
    if not torch.equal(radii_g[0].cpu() > 0, radii64 > 0):
        print("  cull decisions differ; skipped")
        return
    tw, th = (W + 15) // 16, (H + 15) // 16
    radii_use = radii_g[0].cpu()
    _, iids, fids = G.isect_tiles(m2_g[0].detach().cpu(), radii_use, dep_g[0].detach().cpu(), 16, tw, th)
    offs = G.isect_offset_encode(iids, tw, th)
    if fids.numel() == 0:
        print("  nothing on screen")
        return

    # ---- compositor alone: inputs = the fp64 oracle's projection rounded to fp32, the same for all three ----
    base = dict(m2=m2_64.detach().float(), con=con64.detach().float(), opac=sc["opacities"].float(),
                col=torch.cat([sc["colors"].float(), dep64.detach().float()[:, None]], -1) if "+" in mode else
                (sc["colors"].float() if mode == "RGB" else dep64.detach().float()[:, None]))

    def oracle_raster(dt):
        x = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in base.items()}
        res = G.rasterize_to_pixels(x["m2"], x["con"], x["col"], x["opac"], W, H, 16, offs, fids, None, dt == torch.float64)
        return x, res

    x64, res64 = oracle_raster(torch.float64)
    x32, res32 = oracle_raster(torch.float32)
    stable = ~res64[3]
    g = torch.Generator().manual_seed(seed)
    wt = torch.randn(res64[0].shape, generator=g, dtype=torch.float64) * stable[..., None]
    wa = torch.randn(res64[1].shape, generator=g, dtype=torch.float64) * stable[..., None]
    ((res64[0] * wt).sum() + (res64[1] * wa).sum()).backward()
    ((res32[0] * wt.float()).sum() + (res32[1] * wa.float()).sum()).backward()
    xg = {k: v.detach().clone().cuda()[None].requires_grad_(True) for k, v in base.items()}
    r, a = ops.rasterize_to_pixels(xg["m2"], xg["con"], xg["col"], xg["opac"], W, H, 16, offs.cuda()[None].contiguous(), fids.cuda(), None, True)
    ((r[0] * wt.float().cuda()).sum() + (a[0] * wa.float().cuda()).sum()).backward()
    print("  compositor backward alone (norm-rel, worst element, 99th pct):")
    for k in base:
        print(f"    {k:5s} HIP {fmt(grad_errors(xg[k].grad[0], x64[k].grad))}   oracle fp32 {fmt(grad_errors(x32[k].grad, x64[k].grad))}")

    # ---- projection backward alone: incoming gradients = the fp64 compositor's, rounded to fp32 ----
    v_m2, v_con = x64["m2"].grad.float(), x64["con"].grad.float()
    v_dep = x64["col"].grad[:, -1].float() if mode != "RGB" else torch.zeros(N)

    def oracle_proj_bwd(inp, m2, dep, con, dt):
        return torch.autograd.grad([m2, con, dep], [inp[k] for k in NAMES], [v_m2.to(dt), v_con.to(dt), v_dep.to(dt)], allow_unused=True)

    g64 = oracle_proj_bwd(in64, m2_64, dep64, con64, torch.float64)
    g32 = oracle_proj_bwd(in32, m2_32, dep32, con32, torch.float32)
    gg = torch.autograd.grad([m2_g, con_g, dep_g], [gin[k] for k in NAMES], [v_m2.cuda()[None], v_con.cuda()[None], v_dep.cuda()[None]])
    print("  projection backward alone:")
    for i, k in enumerate(NAMES):
        print(f"    {k:6s} HIP {fmt(grad_errors(gg[i], g64[i]))}   oracle fp32 {fmt(grad_errors(g32[i], g64[i]))}")
    # worst quaternion row
    e = ((gg[1].cpu().double() - g64[1]).abs() / g64[1].abs().max()).max(dim=1)
    w = int(e.values.argmax())
    print(f"    worst quat row {w}: |q| {float(sc['quats'][w].norm()):.3g} scales {sc['scales'][w].tolist()} radius {int(radii_use[w])}")
    print(f"      HIP  {gg[1][w].tolist()}\n      fp64 {g64[1][w].tolist()}\n      fp32 {g32[1][w].tolist()}")


if __name__ == "__main__":
    for s in sys.argv[1:]:
        main(int(s))
