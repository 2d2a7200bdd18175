#!/usr/bin/env python3
"""The reference's loop shape (tools/train.py:250-283: a random image per step) on ONE captured view: graph_view.FrameGraph(dynamic=True)
replays it with another camera / target / image index every step, FusedAdam(consume_grads=True) clears every gradient as it consumes
it, step() returns the frame's validity.  Fits six 640x360 targets rendered from a ground-truth scene (L1 + TV loss, the loss the
replayed step carries); prints PSNR and the step time.  Run on the GPU box:  python scripts/train_loop_graph_demo.py [steps]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import harness as Hn
from bilateral_driving_amd.graph_view import FrameGraph
from bilateral_driving_amd.optim import DeferredRowAdam, FusedAdam

dev = torch.device("cuda", 0)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
DEFERRED = len(sys.argv) > 2 and sys.argv[2] == "deferred"   # the SH rows step row-lazily (optim.DeferredRowAdam): the same numbers
W, H, N_GT = 640, 360, 60_000
torch.manual_seed(0)
cams = Hn.ring_cameras(W, H, device=dev)
gt = Hn.synthetic_scene(N_GT, seed=1, device=dev)
gt_grids = Hn.make_grids(len(cams), seed=3, device=dev)
sky = torch.rand(H, W, 3, device=dev)
with torch.no_grad():
    targets = [Hn.render_view(gt, c, gt_grids, v, sky)["rgb"].clone() for v, c in enumerate(cams)]
sel = torch.randperm(N_GT, device=dev)[: N_GT // 2]
p = {"means": gt["means"][sel] + 0.05 * torch.randn(len(sel), 3, device=dev), "quats": gt["quats"][sel].clone(),
     "log_scales": gt["log_scales"][sel] + 0.2, "opacity_logits": torch.full((len(sel),), -1.0, device=dev),
     "sh": torch.zeros(len(sel), 16, 3, device=dev)}
p = {k: v.contiguous().requires_grad_(True) for k, v in p.items()}
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), seed=0, device=dev)]
lrs = dict(means=1.6e-3, quats=1e-3, log_scales=5e-3, opacity_logits=5e-2, sh=2.5e-3)
groups = [{"params": [p[k]], "lr": lr, "eps": 1e-15} for k, lr in lrs.items() if not (DEFERRED and k == "sh")] + [{"params": [g], "lr": 2e-3, "eps": 1e-15} for g in grids]
if DEFERRED:
    groups.append({"params": [p["sh"]], "lr": lrs["sh"], "eps": 1e-15, "deferred_rows": True})
opt = (DeferredRowAdam if DEFERRED else FusedAdam)(groups, lr=0.0, eps=1e-15, consume_grads=True)
frame = FrameGraph(p, [cams[0]], grids, [sky], [targets[0]], img_indices=[0], dynamic=True, calib_cams=cams, clear_grads=False,
                   row_catchup=opt.catchup if DEFERRED else None)


def psnr():
    if DEFERRED:
        opt.flush()     # (the evaluation below reads the SH rows of every Gaussian through another path)
    with torch.no_grad():
        mse = sum(float(((Hn.render_view(p, c, grids, v, sky)["rgb"] - targets[v]) ** 2).mean()) for v, c in enumerate(cams))
    return -10 * math.log10(mse / len(cams))


print(f"step {0:4d}  PSNR {psnr():6.2f} dB")
torch.cuda.synchronize(); t0 = time.perf_counter(); skipped = 0
for step in range(1, STEPS + 1):
    v = int(torch.randint(0, len(cams), (1,)))
    frame.set_view(0, cams[v], targets[v], sky, v)
    if frame.step():
        opt.step(lists=frame.row_lists()) if DEFERRED else opt.step()
    else:
        skipped += 1
    if step % 200 == 0:
        torch.cuda.synchronize()
        print(f"step {step:4d}  PSNR {psnr():6.2f} dB  ({(time.perf_counter() - t0) / step * 1e3:.2f} ms/step incl. evaluation, {skipped} frames repeated, "
              f"{frame.n_captures} captures)")
