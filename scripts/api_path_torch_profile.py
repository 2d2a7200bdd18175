"""Per-operator GPU time of the drop-in call sequence (torch profiler): `python scripts/api_path_torch_profile.py [eager|installed]`.
eager: harness.render_view_api (dense activations + dense SH); installed: the class's get_gaussians through marshalling.install."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bilateral_driving_amd import harness as Hn
from bilateral_driving_amd import marshalling as M
mode = sys.argv[1] if len(sys.argv) > 1 else "installed"
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
for c in cams: c.viewmat.requires_grad_(True)
params = Hn.synthetic_scene(N, seed=0, device=dev)
for v in params.values(): v.requires_grad_(True)
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
gen = torch.Generator().manual_seed(7)
sky = torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True); target = torch.rand(H, W, 3, generator=gen).to(dev)
if mode == "installed":
    model = Hn.VanillaModel(params)
    M.install(Hn.VanillaModel)
    def step(i):
        v = i % len(cams)
        for p in model.parameters() + grids + [sky, cams[v].viewmat]: p.grad = None
        Hn.training_loss(Hn.render_view_model(model, cams[v], grids, v, sky), target, grids).backward()
else:
    Hn.FUSED = False
    def step(i):
        v = i % len(cams)
        for p in list(params.values()) + grids + [sky, cams[v].viewmat]: p.grad = None
        Hn.training_loss(Hn.render_view(params, cams[v], grids, v, sky), target, grids).backward()
for i in range(6): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12): step(i)
torch.cuda.synchronize()
print(f"# {mode}: {12 / (time.perf_counter() - t0):.1f} views/s")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(6): step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=70))
