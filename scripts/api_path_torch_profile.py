import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bilateral_driving_amd import harness as Hn
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
params = Hn.synthetic_scene(N, seed=0, device=dev)
for v in params.values(): v.requires_grad_(True)
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
sky = torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True); target = torch.rand(H, W, 3, generator=gen).to(dev)
Hn.FUSED = False
def step(i):
    v = i % len(cams)
    for p in list(params.values()) + grids: p.grad = None
    Hn.training_loss(Hn.render_view(params, cams[v], grids, v, sky), target, grids).backward()
for i in range(4): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(2): step(i)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40, max_src_column_width=110))
