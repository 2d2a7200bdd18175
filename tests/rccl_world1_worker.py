"""Worker of tests/test_gpu_24_rccl_world1.py (run as a child process under a timeout: a hung collective must not hang pytest).

ONE rank, backend "nccl" (= RCCL), ``dist.force_collectives(True)``: every collective of the multi-GPU exchange is really issued --
uint8 MAX all-reduce of the visibility mask, asynchronous SUM all-reduce of the compact rows between a view's captured graphs, the
dense per-frame all-reduce, the flags of ``valid()``, the capture's warm-up frame next to RCCL's proxy thread.  A one-rank all-reduce
is the identity: the frame's gradients must equal the no-exchange frame's (up to the run-to-run noise of the float atomics in the
composite backward's gradient records, which the worker measures on the no-exchange frame itself).  Prints one JSON line.
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, N = 256, 160, 6000
YAWS = (0.0, 100.0, 200.0)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == 1
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from bilateral_driving_amd import dist as D
    from bilateral_driving_amd import fused_view as FV
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    from bilateral_driving_amd.graph_view import FrameGraph
    FV.SH_IN_PACK = FV.SH_IN_PACK_DEV      # (same SH arithmetic in the eager and the captured form)
    D.force_collectives(True)
    assert D._active()
    out = {"backend": dist.get_backend(), "world": world, "modes": {}}

    cams = Hn.ring_cameras(W, H, yaws_deg=YAWS, device=dev)
    base = Hn.synthetic_scene(N, seed=4, device=dev)
    grids0 = Hn.make_grids(5, device=dev)
    g = torch.Generator().manual_seed(9)
    skies = [torch.rand(H, W, 3, generator=g).to(dev) for _ in cams]
    targets = [torch.rand(H, W, 3, generator=g).to(dev) for _ in cams]
    V = len(cams)

    def leaves():
        p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
        return p, [x.clone().requires_grad_(True) for x in grids0]

    def grads(p, grids):
        return torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids]).clone()

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    # ---- the no-exchange frame (and its own run-to-run noise: float atomics in the gradient records) ----
    p, grids = leaves()
    plain = FrameGraph(p, cams, grids, [s.clone() for s in skies], targets)
    refs = []
    for _ in range(5):
        assert plain.step() is True
        refs.append(grads(p, grids))
    ref = refs[0]
    noise = max(rel(r, ref) for r in refs[1:])
    out["plain_noise"] = noise
    bound = max(20.0 * noise, 1e-6)

    # ---- fixed frames with an exchange: per view (collectives BETWEEN the graphs) and per frame (one dense all-reduce) ----
    for per_view in (True, False):
        for overlap in (True, False):
            p, grids = leaves()
            flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
            fx = FrameExchange(flat, list(p.keys()) + [f"grid{i}" for i in range(len(grids))], per_view=per_view)
            assert fx.coll and fx.world == 1 and fx.active == per_view and fx.frame_reduce == (not per_view)
            n0 = dict(D.ISSUED)
            frame = FrameGraph(p, cams, grids, [s.clone() for s in skies], targets, exchange=fx, overlap=overlap)
            n_build = D.ISSUED["all_reduce"] - n0["all_reduce"]
            worst, n1 = 0.0, D.ISSUED["all_reduce"]
            for _ in range(5):
                assert frame.step() is True
                worst = max(worst, rel(grads(p, grids), ref))
            per_frame = (D.ISSUED["all_reduce"] - n1) / 5.0
            # per view: mask MAX + rows SUM per view, the tail SUM, the flags of valid(); per frame: the dense SUM + the flags
            want = (2 * V + 2) if per_view else 2
            assert per_frame == want, (per_view, per_frame, want)
            assert worst <= bound, (per_view, overlap, worst, bound)
            n2 = D.ISSUED["all_reduce"]
            frame.recapture()                       # per view a collective: calibration masks + a warm-up frame of its own
            assert (D.ISSUED["all_reduce"] > n2) == per_view
            assert frame.step() is True
            again = rel(grads(p, grids), ref)
            assert again <= bound, (per_view, overlap, again, bound)
            out["modes"][f"fixed per_view={per_view} overlap={overlap}"] = {
                "worst_rel_vs_plain": worst, "after_recapture": again, "all_reduces_per_frame": per_frame, "all_reduces_in_build": n_build,
                "payload_bytes_per_frame": fx.payload_bytes, "captures": frame.n_captures}
            del frame, fx, flat

    # ---- replayable slots (dynamic=True) with an exchange ----
    pool = []
    g2 = torch.Generator().manual_seed(11)
    for k in range(3):
        pool += Hn.ring_cameras(W, H, yaws_deg=[y + float(torch.rand(1, generator=g2)) * 30.0 - 15.0 for y in YAWS], device=dev,
                                origin=(0.4 * k, 0.0, 0.0))
    for per_view in (True, False):
        p, grids = leaves()
        flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
        fx = FrameExchange(flat, list(p.keys()) + [f"grid{i}" for i in range(len(grids))], per_view=per_view)
        frame = FrameGraph(p, pool[:2], grids, [skies[0].clone() for _ in range(2)], [targets[0].clone() for _ in range(2)], exchange=fx,
                           dynamic=True, calib_cams=pool, img_indices=[0, 0])
        q, qg = leaves()
        plain_dyn = FrameGraph(q, pool[:2], qg, [skies[0].clone() for _ in range(2)], [targets[0].clone() for _ in range(2)],
                               dynamic=True, calib_cams=pool, img_indices=[0, 0])
        worst = 0.0
        for step in range(5):
            picks = [(int(torch.randint(0, len(pool), (1,), generator=g2)), int(torch.randint(0, V, (1,), generator=g2)),
                      int(torch.randint(0, 5, (1,), generator=g2))) for _ in range(2)]
            for fr in (frame, plain_dyn):
                for v, (c, t, i) in enumerate(picks):
                    fr.set_view(v, pool[c], targets[t], skies[(t + 1) % V], i)
                assert fr.step() is True
            worst = max(worst, rel(grads(p, grids), grads(q, qg)))
        assert worst <= bound, ("dynamic", per_view, worst, bound)
        assert frame.n_captures == 1
        out["modes"][f"dynamic per_view={per_view}"] = {"worst_rel_vs_plain": worst, "captures": frame.n_captures}
        del frame, plain_dyn, fx, flat

    # ---- the eager frame loop with the per-view exchange (rotating buffers, deferred overflow check) ----
    p, grids = leaves()
    flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
    fx = FrameExchange(flat, list(p.keys()) + [f"grid{i}" for i in range(len(grids))], per_view=True)
    worst = 0.0
    for _ in range(3):
        fx.begin_frame()
        for v, cam in enumerate(cams):
            o = Hn.render_view(p, cam, grids, v, skies[v], **fx.view_kwargs(v))
            fx.begin_view(o["info"])
            Hn.training_loss(o, targets[v], grids).backward()
            fx.end_view()
        fx.end_frame()
        worst = max(worst, rel(grads(p, grids), ref))
    assert worst <= max(bound, 1e-4), ("eager", worst)        # (the eager loss launch sums in another order than the fused one)
    out["modes"]["eager per_view=True"] = {"worst_rel_vs_plain": worst}

    # ---- the small collectives: densification statistics, rank 0's noise, FlatGradients' own row exchange ----
    a, b, c = torch.rand(N, device=dev), torch.rand(N, device=dev), torch.rand(N, device=dev)
    a0, b0, c0 = a.clone(), b.clone(), c.clone()
    D.reduce_densify_stats(a, b, c)
    assert torch.equal(a, a0) and torch.equal(b, b0) and torch.equal(c, c0)
    z = D.broadcast_randn((7, 3), dev)
    assert z.shape == (7, 3) and bool(torch.isfinite(z).all())
    p, grids = leaves()
    flat = FlatGradients(list(p.values()) + grids)
    for t in list(p.values()) + grids:
        t.grad = torch.rand_like(t)
    want = grads(p, grids)
    touched = torch.rand(N, device=dev) < 0.3
    for t in p.values():
        t.grad[~touched] = 0
    want = grads(p, grids)
    flat.begin_rows_union(touched)
    flat.all_reduce()
    assert torch.equal(grads(p, grids), want)
    flat.all_reduce(async_op=True)
    flat.wait()
    assert torch.equal(grads(p, grids), want)
    bw = D.measure_busbw(dev, nbytes=64 << 20)
    out["busbw_world1"] = bw
    out["issued"] = dict(D.ISSUED)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1 " + json.dumps(out))


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"done in {time.time() - t0:.1f} s")
