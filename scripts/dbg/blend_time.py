"""fused view backward's bilateral stage alone at the headline shape through harness (ED form with sky): per-operator event times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bilateral_driving_amd import _lib as L, harness as Hn
from bilateral_driving_amd.graph_view import FrameGraph
dev = torch.device("cuda", 0); N, W, H = 2_000_000, 1920, 1080
cams = Hn.ring_cameras(W, H, device=dev)
for c in cams: c.viewmat.requires_grad_(True)
p0 = Hn.synthetic_scene(N, seed=0, device=dev); perm = Hn.spatial_order(p0["means"])
p = {k: v[perm].contiguous().requires_grad_(True) for k, v in p0.items()}
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
L.enable_timers(True)
fr = FrameGraph(p, cams, grids, skies, targets)
L.enable_timers(False)
acc = {}
for _ in range(5):
    fr.step(serial=True); torch.cuda.synchronize()
    for n in fr.marks: acc.setdefault(n, []).extend(fr.mark_samples(n))
print({k: round(sum(v) / len(v) * 1e3, 1) for k, v in sorted(acc.items())})
import time
for _ in range(3): fr.step(wait=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): fr.step(wait=False)
torch.cuda.synchronize(); print("it/s", 6 * 40 / (time.perf_counter() - t0))
