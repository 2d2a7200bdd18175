"""What the multi-GPU exchange costs ONE rank (RCCL at world size 1, dist.force_collectives): ms per frame and host enqueue time of the
frame without an exchange, the compact path without a collective, per view over RCCL (two streams / one), per frame."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29566")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from bilateral_driving_amd import dist as D, harness as Hn
from bilateral_driving_amd.graph_view import FrameGraph
N, W, H = 2_000_000, 1920, 1080
cams = Hn.ring_cameras(W, H, device=dev)
for c in cams: c.viewmat.requires_grad_(True)
params = Hn.synthetic_scene(N, seed=0, device=dev)
perm = Hn.spatial_order(params["means"]); params = {k: v[perm].contiguous().requires_grad_(True) for k, v in params.items()}
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
names = list(params.keys()) + [f"grid{i}" for i in range(len(grids))]
def timed(frame, n=20):
    for _ in range(3): assert frame.step() is True
    torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        h0 = time.perf_counter(); frame.step(wait=False); host += time.perf_counter() - h0
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    return round(dt, 3), round(host / n * 1e3, 3)
print("plain (ms/frame, host enqueue ms/frame):", timed(FrameGraph(params, cams, grids, skies, targets)))
for label, force_coll, force, pv, ov in (("compact path, no collective", False, True, True, True), ("per view + RCCL", True, False, True, True),
                                         ("per view + RCCL, one stream", True, False, True, False), ("per frame + RCCL", True, False, False, True)):
    D.force_collectives(force_coll)
    flat = D.FlatGradients(list(params.values()) + grids, sparse_rows=True)
    fx = D.FrameExchange(flat, names, per_view=pv, force=force)
    fr = FrameGraph(params, cams, grids, skies, targets, exchange=fx, overlap=ov)
    print(label, timed(fr), "cap", fx.cap)
    del fr, fx, flat
dist.destroy_process_group()
