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
    if rank == 0:
        print("[dist_check]", "PASS" if ok else "FAIL")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
