"""One-view training steps with optim.DeferredRowAdam, one stream (`one`) or -- with profiles/r09j_split_forward_side_stream.patch applied --
the SH step on the optimizer's own stream behind a split forward (`side`): ms per step + host enqueue time (profiles/NOTES.md, round 6)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bilateral_driving_amd import harness as Hn
from bilateral_driving_amd.graph_view import FrameGraph
from bilateral_driving_amd.optim import DeferredRowAdam
side = len(sys.argv) > 1 and sys.argv[1] == "side"
dev = torch.device("cuda", 0); N, W, H = 2_000_000, 1920, 1080
cams = Hn.ring_cameras(W, H, device=dev)
for c in cams: c.viewmat.requires_grad_(True)
params = Hn.synthetic_scene(N, seed=0, device=dev)
perm = Hn.spatial_order(params["means"]); p = {k: v[perm].contiguous().requires_grad_(True) for k, v in params.items()}
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
lrs = dict(means=1.6e-4, quats=1e-3, log_scales=5e-3, opacity_logits=5e-2)
groups = [{"params": [p[k]], "lr": lr, "eps": 1e-15} for k, lr in lrs.items()] + [{"params": [x], "lr": 2e-3, "eps": 1e-15} for x in grids]
groups.append({"params": [p["sh"]], "lr": 2.5e-3, "lr_b": 1.25e-4, "col_split": 3, "deferred_rows": True, "eps": 1e-15})
opt = DeferredRowAdam(groups, lr=0.0, eps=1e-15, consume_grads=True, **({"side_stream": True} if side else {}))
fr = FrameGraph(p, cams[:1], grids, skies[:1], targets[:1], img_indices=[0], dynamic=True, calib_cams=cams, clear_grads=False,
                row_catchup=opt.catchup, **({"split_forward": True, "row_sync": opt.row_sync} if side else {}))
def one(i):
    k = i % len(cams)
    fr.set_view(0, cams[k], targets[k], skies[k].detach(), k)
    fr.step(wait=False)
    opt.step(lists=fr.row_lists())
for i in range(5): one(i)
torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0
for i in range(20):
    h0 = time.perf_counter(); one(i); host += time.perf_counter() - h0
torch.cuda.synchronize()
print("side" if side else "one", "ms/step", (time.perf_counter() - t0) / 20 * 1e3, "host ms/step", host / 20 * 1e3)
