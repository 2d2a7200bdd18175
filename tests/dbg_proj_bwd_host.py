"""CPU: the projection backward's HD math (tests/hostmath_shim.hip) against the oracle's autograd in fp64 / fp32 on one sweep scene.
    python tests/dbg_proj_bwd_host.py <seed>"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gs_oracle as G
from tests.test_gpu_25_gs_random_sweep import random_scene
from tests.util import grad_errors, hostmath, fptr

NAMES = ("means", "quats", "scales")
seed = int(sys.argv[1])
sc, W, H, mode, kw, bg = random_scene(seed)
N = sc["means"].shape[0]
vm, K = sc["viewmats"], sc["Ks"]

def proj(dt):
    inp = {k: sc[k].detach().clone().to(dt).requires_grad_(True) for k in NAMES}
    radii, m2, dep, con, _ = G.project(inp["means"], inp["quats"], inp["scales"], vm[0].to(dt), K[0].to(dt), W, H, 0.3, kw["near_plane"], 1e10, kw["radius_clip"])
    return inp, radii, m2, dep, con
in64, radii, m2, dep, con = proj(torch.float64)
in32, radii32, m2_32, dep32, con32 = proj(torch.float32)
# incoming gradients: the fp64 compositor's on this scene (as tests/dbg_gs_stage_errors.py)
tw, th = (W + 15) // 16, (H + 15) // 16
_, iids, fids = G.isect_tiles(m2.detach(), radii, dep.detach(), 16, tw, th)
offs = G.isect_offset_encode(iids, tw, th)
col = torch.cat([sc["colors"].double(), dep.detach()[:, None]], -1) if "+" in mode else (sc["colors"].double() if mode == "RGB" else dep.detach()[:, None])
x = dict(m2=m2.detach().clone().requires_grad_(True), con=con.detach().clone().requires_grad_(True), col=col.clone().requires_grad_(True))
res = G.rasterize_to_pixels(x["m2"], x["con"], x["col"], sc["opacities"].double(), W, H, 16, offs, fids, None, True)
stable = ~res[3]
g = torch.Generator().manual_seed(seed)
wt = torch.randn(res[0].shape, generator=g, dtype=torch.float64) * stable[..., None]
wa = torch.randn(res[1].shape, generator=g, dtype=torch.float64) * stable[..., None]
((res[0] * wt).sum() + (res[1] * wa).sum()).backward()
v_m2, v_con = x["m2"].grad.float(), x["con"].grad.float()
v_dep = x["col"].grad[:, -1].float() if mode != "RGB" else torch.zeros(N)
g64 = torch.autograd.grad([m2, con, dep], [in64[k] for k in NAMES], [v_m2.double(), v_con.double(), v_dep.double()])
g32 = torch.autograd.grad([m2_32, con32, dep32], [in32[k] for k in NAMES], [v_m2, v_con, v_dep])
hm = hostmath()
A = lambda t: np.ascontiguousarray(t.detach().numpy().astype(np.float32))
means, quats, scales = A(sc["means"]), A(sc["quats"]), A(sc["scales"])
vmn, Kn = A(vm[0]), A(K[0])
rad = np.ascontiguousarray(radii.numpy().astype(np.int32))
o_m, o_q, o_s = np.zeros((N, 3), np.float32), np.zeros((N, 4), np.float32), np.zeros((N, 3), np.float32)
vR, vt = np.zeros(9, np.float32), np.zeros(3, np.float32)
import ctypes
hm.hm_project_bwd(N, fptr(means), fptr(quats), fptr(scales), fptr(vmn), fptr(Kn), W, H, ctypes.c_float(0.3), fptr(rad), fptr(A(v_m2)), fptr(A(v_dep)), fptr(A(v_con)),
                  fptr(o_m), fptr(o_q), fptr(o_s), fptr(vR), fptr(vt))
for i, (k, o) in enumerate(zip(NAMES, (o_m, o_q, o_s))):
    print(f"{k:6s} host-shim {tuple(f'{x:.1e}' for x in grad_errors(torch.from_numpy(o), g64[i]))}  oracle fp32 {tuple(f'{x:.1e}' for x in grad_errors(g32[i], g64[i]))}")
# per-row relative error of the quaternion gradient
e = (torch.from_numpy(o_q).double() - g64[1]).norm(dim=1) / g64[1].norm(dim=1).clamp(min=1e-30)
e32 = (g32[1].double() - g64[1]).norm(dim=1) / g64[1].norm(dim=1).clamp(min=1e-30)
ratio = sc["scales"].max(dim=1).values / sc["scales"].min(dim=1).values
idx = torch.argsort(e, descending=True)[:8]
for i in idx.tolist():
    print("   HIPmath", o_q[i].tolist(), "\n   fp64   ", g64[1][i].tolist(), "\n   fp32   ", g32[1][i].tolist(), "\n   v_con", v_con[i].tolist(), "con", con[i].tolist(), "scales", sc["scales"][i].tolist())
    print(f"row {i}: err {float(e[i]):.2e} (fp32 oracle {float(e32[i]):.2e}) aniso {float(ratio[i]):.0f} |q| {float(sc['quats'][i].norm()):.2f} radius {int(radii[i])}")
