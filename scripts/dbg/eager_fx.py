"""The eager per-view exchange loop over RCCL at world size 1 against the dense sum, frame by frame (found the stale id-list bug of
round 6: third frame wrong before the fix in dist.FrameExchange._retire)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from bilateral_driving_amd import dist as D, fused_view as FV, harness as Hn
from bilateral_driving_amd.dist import FlatGradients, FrameExchange
from bilateral_driving_amd.graph_view import FrameGraph
FV.SH_IN_PACK = FV.SH_IN_PACK_DEV
W, H, N = 256, 160, 6000
cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 100.0, 200.0), device=dev)
base = Hn.synthetic_scene(N, seed=4, device=dev)
grids0 = Hn.make_grids(5, device=dev)
g = torch.Generator().manual_seed(9)
skies = [torch.rand(H, W, 3, generator=g).to(dev) for _ in cams]
targets = [torch.rand(H, W, 3, generator=g).to(dev) for _ in cams]
def leaves():
    p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
    return p, [x.clone().requires_grad_(True) for x in grids0]
def grads(p, grids): return torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids]).clone()
def rel(a, b): return float((a - b).norm() / b.norm())
p, grids = leaves()
plain = FrameGraph(p, cams, grids, [s.clone() for s in skies], targets)
plain.step(); ref = grads(p, grids)
# dense reference
refd = None
for v, cam in enumerate(cams):
    q, qg = leaves()
    Hn.training_loss(Hn.render_view(q, cam, qg, v, skies[v]), targets[v], qg).backward()
    x = grads(q, qg); refd = x if refd is None else refd + x
print("plain graph vs dense", rel(ref, refd))
sizes = [t.numel() for t in list(p.values()) + grids]
def parts(a, b):
    o, out = 0, []
    for n in sizes:
        out.append(float((a[o:o+n]-b[o:o+n]).norm() / (b[o:o+n].norm() + 1e-30))); o += n
    return [f"{x:.1e}" for x in out]
for force_coll, force in ((False, True), (True, False)):
    D.force_collectives(force_coll)
    p, grids = leaves()
    flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
    fx = FrameExchange(flat, list(p.keys()) + [f"grid{i}" for i in range(len(grids))], per_view=True, force=force)
    for fr in range(3):
        fx.begin_frame()
        for v, cam in enumerate(cams):
            o = Hn.render_view(p, cam, grids, v, skies[v], **fx.view_kwargs(v))
            fx.begin_view(o["info"])
            Hn.training_loss(o, targets[v], grids).backward()
            fx.end_view()
        fx.end_frame()
        torch.cuda.synchronize()
        got = grads(p, grids)
        print(f"coll={force_coll} frame {fr}: vs dense {rel(got, refd):.2e} vs graph {rel(got, ref):.2e}", parts(got, refd))
dist.destroy_process_group()
