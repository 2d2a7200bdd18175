#!/usr/bin/env python3
"""Plumbing check of the multi-GPU exchange step on a box with ONE GPU: two processes share cuda:0 and talk
through gloo (RCCL refuses two ranks on one device), running the same code path bench.py uses at N > 1
(fused view with gradient arena -> FlatGradients.all_reduce).  Rank 0 verifies
    all-reduced gradient == sum of the two views' gradients computed sequentially on one process.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 scripts/dist_check.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bilateral_driving_amd import harness as Hn  # noqa: E402
from bilateral_driving_amd.dist import FlatGradients, view_for_rank  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    N, W, H = 200_000, 640, 360
    cams = Hn.ring_cameras(W, H, device=dev)
    base = Hn.synthetic_scene(N, seed=0, device=dev)
    grids0 = Hn.make_grids(len(cams), device=dev)
    gen = torch.Generator().manual_seed(7)
    skies = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
    targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]

    def run_view(v, params, grids, arena, flat=None):
        out = Hn.render_view(params, cams[v], grids, v, skies[v], grad_arena=arena)
        if flat is not None:   # as bench.py: only rows visible on some rank travel
            flat.begin_rows_union(out["info"]["radii"][0] > 0)
        Hn.training_loss(out, targets[v], grids).backward()

    params = {k: t.clone().requires_grad_(True) for k, t in base.items()}
    grids = [g.clone().requires_grad_(True) for g in grids0]
    flat = FlatGradients(list(params.values()) + grids)
    arena = flat.arena(list(params.keys()))
    ok = True
    for step in range(4):   # even steps: rows-union exchange, odd steps: dense all-reduce
        flat.zero()
        run_view(view_for_rank(step, rank, world, len(cams)), params, grids, arena, flat if step % 2 == 0 else None)
        flat.all_reduce()
        torch.cuda.synchronize()
        if rank == 0:
            p2 = {k: t.clone().requires_grad_(True) for k, t in base.items()}
            g2 = [g.clone().requires_grad_(True) for g in grids0]
            for r in range(world):  # sequential accumulation of the same views on one process
                run_view(view_for_rank(step, r, world, len(cams)), p2, g2, None)
            ref = torch.cat([t.grad.reshape(-1) for t in list(p2.values()) + g2])
            rel = float((flat.flat - ref).norm() / ref.norm())
            aliased = all(t.grad.data_ptr() == v.data_ptr() for t, v in zip(flat.params, flat._views))
            print(f"[dist_check] step {step}: rel err {rel:.2e}, grads alias the reduced buffer: {aliased}, payload {flat.last_payload_bytes / 1e6:.1f} of {flat.nbytes / 1e6:.1f} MB")
            ok = ok and rel < 1e-3 and aliased
    ok = refinement_phase(rank, world, dev, base) and ok
    if rank == 0:
        print("[dist_check]", "PASS" if ok else "FAIL")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def refinement_phase(rank, world, dev, base):
    """View-parallel densification: per-rank statistics are reduced, rank 0's split noise is broadcast, every replica runs the same
    refinement -> the replicas must end up bit-identical (and equal to one process that saw both ranks' statistics)."""
    import types
    from bilateral_driving_amd.dist import refinement_after_synced
    from bilateral_driving_amd.densify import refinement_after
    from bilateral_driving_amd.optim import FusedAdam
    N = base["means"].shape[0]
    ctrl = types.SimpleNamespace(warmup_steps=500, reset_alpha_interval=3000, refine_interval=100, n_split_samples=2, reset_alpha_value=0.01,
                                 densify_grad_thresh=0.0003, densify_size_thresh=0.002, cull_alpha_thresh=0.005, cull_scale_thresh=0.5,
                                 cull_screen_size=0.15, split_screen_size=0.05, stop_screen_size_at=4000, stop_split_at=15000)

    def make():
        m = types.SimpleNamespace(ctrl_cfg=ctrl, scene_scale=30.0, num_train_images=6, step=3300, class_prefix="Background#")
        P = torch.nn.Parameter
        m._means, m._quats, m._scales = P(base["means"].clone()), P(base["quats"].clone()), P(base["log_scales"].clone())
        m._opacities = P(base["opacity_logits"].clone()[:, None])
        m._features_dc, m._features_rest = P(base["sh"][:, 0].clone()), P(base["sh"][:, 1:].clone())
        names = dict(_means="xyz", _features_dc="sh_dc", _features_rest="sh_rest", _opacities="opacity", _scales="scaling", _quats="rotation")
        opt = FusedAdam([{"params": [getattr(m, a)], "name": m.class_prefix + n, "lr": 1e-3} for a, n in names.items()], lr=0.0, eps=1e-15)
        for a in names:
            getattr(m, a).grad = torch.full_like(getattr(m, a), 1e-3)
        opt.step()
        return m, opt, list(names)

    def rank_stats(r):   # what rank r's views would have accumulated (rank 0 carries the reference's initial one in vis_counts)
        g = torch.Generator().manual_seed(100 + r)
        return (torch.rand(N, generator=g).to(dev) * 0.004, torch.floor(torch.rand(N, generator=g) * 3).to(dev) + (1.0 if r == 0 else 0.0),
                torch.rand(N, generator=g).to(dev) * 0.2)

    m, opt, attrs = make()
    m.xys_grad_norm, m.vis_counts, m.max_2Dsize = rank_stats(rank)
    torch.manual_seed(1234 + rank)                       # different RNG streams on the ranks: the broadcast has to make them agree
    refinement_after_synced(m, 3300, opt)
    sig = torch.stack([getattr(m, a).detach().double().sum() for a in attrs] +
                      [opt.state[getattr(m, a)]["exp_avg_sq"].double().sum() for a in attrs] + [torch.tensor(float(m._means.shape[0]), device=dev, dtype=torch.float64)])
    sigs = [torch.empty_like(sig) for _ in range(world)]
    dist.all_gather(sigs, sig)
    same = all(torch.equal(sigs[0], s) for s in sigs)
    ok = same and m._means.shape[0] != N
    if rank == 0:   # one process with the summed statistics and the same noise stream gives the same set
        m1, opt1, _ = make()
        st = [rank_stats(r) for r in range(world)]
        m1.xys_grad_norm, m1.vis_counts = sum(s[0] for s in st), sum(s[1] for s in st)
        m1.max_2Dsize = torch.stack([s[2] for s in st]).max(dim=0).values
        torch.manual_seed(1234)
        refinement_after(m1, 3300, opt1, verbose=False)
        seq = m1._means.shape == m._means.shape and torch.equal(m1._means, m._means) and torch.equal(m1._features_rest, m._features_rest)
        print(f"[dist_check] synced refinement: {N} -> {m._means.shape[0]} Gaussians, replicas identical: {same}, equals one process: {seq}")
        ok = ok and seq
    return ok


if __name__ == "__main__":
    main()
