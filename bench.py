#!/usr/bin/env python3
"""Benchmark of the hot path: train iterations/sec (forward + backward of one view per GPU per step)
of the differentiable Gaussian-splat rasterizer + fused multi-scale bilateral-grid colour transform.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): 2 M Gaussians, 6-camera ring at 1920x1080, SH degree 3, 3-level
bilateral grid; synthetic scene of SURVEY.md 8(d); one step = SH -> projection -> tile intersection +
ordering -> alpha compositing (RGB + expected depth) -> clamp + sky blend + bilateral slice + affine ->
L1 + TV loss -> full backward (-> one all-reduce of the flat per-Gaussian gradients when N > 1).
At N ranks, rank r renders view (step*N + r) mod 6 ("weak" scaling: one view per GPU per step);
value = views processed by all ranks per second.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the launch stream inside the timed region) and `cpu_baseline` (the oracle/ port of
the same pipeline on the host cores, bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-gaussians", type=int, default=100000)
    ap.add_argument("--cpu-sample-width", type=int, default=960)
    ap.add_argument("--cpu-sample-height", type=int, default=540)
    ap.add_argument("--cpu-threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--cpu-timeout", type=float, default=150.0)
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--dense-grads", action="store_true",
                    help="fresh dense gradient tensors every step (zero fill of all N rows) instead of the persistent flat gradient "
                         "buffer whose rows are cleared / written through the visible-id lists")
    return ap.parse_args()


def _cpu_iter_fn(N, W, H):
    """One fwd+bwd of the oracle/ port (the checker; here only as the timed CPU baseline)."""
    from bilateral_driving_amd import harness as Hn
    from oracle import bilagrid_oracle as BO
    from oracle import gs_oracle as G

    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,))[0]
    p = Hn.synthetic_scene(N, seed=0)
    grids = Hn.make_grids(1)
    gen = torch.Generator().manual_seed(3)
    sky = torch.rand(H, W, 3, generator=gen)
    target = torch.rand(H, W, 3, generator=gen)

    def one():
        q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        g = [x.clone().requires_grad_(True) for x in grids]
        dirs = q["means"].detach() - torch.linalg.inv(cam.viewmat)[:3, 3]
        col = torch.clamp(G.spherical_harmonics(3, dirs, q["sh"]) + 0.5, 0.0, 1.0)
        r, a, _ = G.rasterization(q["means"], q["quats"] / q["quats"].norm(dim=-1, keepdim=True), torch.exp(q["log_scales"]),
                                  torch.sigmoid(q["opacity_logits"]), col, cam.viewmat[None], cam.K[None], W, H,
                                  near_plane=0.1, render_mode="RGB+ED")
        rgb = BO.multiscale_transform([x[0] for x in g], BO.sky_blend(r[0, ..., :3], a[0], sky), list(Hn.FACTORS_3))
        loss = (rgb - target).abs().mean() + 0.01 * BO.multiscale_tv(g)
        loss.backward()

    return one


def cpu_baseline_worker(args):
    """Runs in a child process (so that a pathological host cannot hang the benchmark): times the
    oracle/ port on a bounded sample and prints one JSON object."""
    threads = args.cpu_threads
    torch.set_num_threads(threads)
    small = (20000, 480, 270)
    big = (args.cpu_sample_gaussians, args.cpu_sample_width, args.cpu_sample_height)
    one = _cpu_iter_fn(*small)
    one()  # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    one()
    t_small = time.perf_counter() - t0
    sample, dt, reps = small, t_small, 1
    # the big sample costs ~15x the small one (measured); only run it if it fits the 10-30 s budget
    if t_small * 15.0 < 35.0:
        one = _cpu_iter_fn(*big)
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        sample = big
    print(json.dumps({
        "value": 1.0 / dt, "unit": "iters/sec", "cores": threads, "kind": "port",
        "sample": f"{sample[0]} Gaussians, one {sample[1]}x{sample[2]} view, SH3 + 3-level bilateral grid, fwd+bwd, fp32 "
                  f"pure-PyTorch CPU port (oracle/), {reps} rep; NOT the GPU workload size "
                  f"(host has {os.cpu_count()} logical CPUs, {threads} torch threads used)",
    }))


def cpu_baseline(args):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-threads", str(args.cpu_threads),
           "--cpu-sample-gaussians", str(args.cpu_sample_gaussians), "--cpu-sample-width", str(args.cpu_sample_width),
           "--cpu-sample-height", str(args.cpu_sample_height)]
    env = dict(os.environ, OMP_NUM_THREADS=str(args.cpu_threads), MKL_NUM_THREADS=str(args.cpu_threads))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.cpu_timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout or failure: report it, never hang the benchmark
        return {"value": None, "unit": "iters/sec", "cores": args.cpu_threads, "kind": "port",
                "sample": f"not measured: {type(e).__name__} (limit {args.cpu_timeout}s)"}


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc.json: separate FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950-corrected).
    bench.py cannot collect counters itself; null when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))["kernels"]
        except Exception:
            continue
        for k, v in d.items():
            if kernel_substr in k:
                return v.get("hbm_bytes_per_launch_corrected"), os.path.basename(f)
    return None, None


def main():
    args = parse()
    if args.cpu_baseline_only:
        cpu_baseline_worker(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, torch.distributed.run on 127.0.0.1)
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU path in the product)"
    # plumbing check on a ONE-GPU box only (scripts / CI): BDS_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and talks through
    # gloo (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    share = os.environ.get("BDS_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run for N>1)"

    from bilateral_driving_amd import _lib as L
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients, view_for_rank

    L.lib()  # fail loudly if libbds.so is missing
    N, W, H = args.gaussians, args.width, args.height
    cams = Hn.ring_cameras(W, H, device=dev)
    for cam in cams:   # the camera pose is learnable in the reference (trainers/base.py:328-329,399): its gradient stays live
        cam.viewmat.requires_grad_(True)
    params = Hn.synthetic_scene(N, seed=0, device=dev)
    for v in params.values():
        v.requires_grad_(True)
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
    gen = torch.Generator().manual_seed(7)
    # the sky colour comes from a trainable sky model in the reference: its gradient path stays live
    skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
    targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
    sparse = not args.dense_grads
    # param.grad = slices of ONE flat buffer (the all-reduce buffer at N > 1): the backward kernels write the rows of the Gaussians a
    # view sees straight into it, zero_grad clears exactly the rows the previous step wrote (a view sees ~15 % of the scene)
    flat = FlatGradients(list(params.values()) + grids, sparse_rows=sparse)
    arena = flat.arena(list(params.keys())) if (world > 1 or sparse) else None

    stats = {}

    def step(s):
        v = view_for_rank(s, rank, world, len(cams))
        flat.zero()
        skies[v].grad = None
        cams[v].viewmat.grad = None
        out = Hn.render_view(params, cams[v], grids, v, skies[v], grad_arena=arena, arena_rows=1 if flat.rows_clean else 0)
        if world > 1:   # rows that can receive a gradient on this rank; the OR over the ranks runs behind the backward pass
            flat.begin_rows_union(out["info"]["radii"][0] > 0)
        elif sparse:
            flat.mark_list(out["info"]["visible_ids"])
        loss = Hn.training_loss(out, targets[v], grids)
        loss.backward()
        flat.all_reduce()
        info = out["info"]
        stats["M"] = info["n_isects"]
        stats["n_visible"] = info["radii"]
        stats["last_ids"] = None
        return loss

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # inside the timed region only the dominant kernel is bracketed by HIP events (every pair is two more packets on the stream);
    # the per-operator table is taken from a few extra, untimed steps afterwards
    L.enable_timers(True, only=("rasterize_bwd",))
    t0 = time.perf_counter()
    Ms = []
    for s in range(args.warmup, args.warmup + args.steps):
        step(s)
        Ms.append(stats["M"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tsum = L.timer_summary()
    L.enable_timers(True)
    for s in range(args.warmup + args.steps, args.warmup + args.steps + 6):
        step(s)
    torch.cuda.synchronize()
    tall = L.timer_summary()
    L.enable_timers(False)
    n_vis = int((stats["n_visible"] > 0).sum())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed

    # roofline of the dominant kernel: composite backward (K8).  Algorithmic bytes per launch
    # (SURVEY.md 8d): 44 B/isect read + 28 B/pixel read + 48 B/isect gradient write.
    M_mean = sum(Ms) / len(Ms)
    dom = "rasterize_bwd"
    calls, mean_ms = tsum.get(dom, (0, float("nan")))
    alg_bytes = 92.0 * M_mean + 28.0 * W * H
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9 if calls else float("nan")
    traffic, traffic_src = pmc_traffic("rasterize_bwd_wave_kernel<4, true>")
    roofline = {"bound": "hbm", "kernel": "bds::rasterize_bwd_wave_kernel<4, true>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": mean_ms,
                "note": "achieved = algorithmic bytes (92 B/isect + 28 B/pixel, SURVEY.md 8d) / HIP-event launch time; the composite "
                        "kernels are VALU-bound (SURVEY.md 7, hard part 2) and L2 serves most re-reads, so HBM traffic << algorithmic bytes"}

    result = {
        "metric": "train iters/sec (fwd+bwd) at 2M Gaussians, 6x1920x1080; HBM roofline %",
        "value": value, "unit": "iters/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{N} Gaussians, 6-cam ring {W}x{H}, SH deg 3, RGB+ED, 3-level bilateral grid "
                               f"[[2,2,1],[4,4,2],[8,8,4]] factors [4,4,2], L1+TV loss, one view per GPU per step",
                   "gaussians": N, "width": W, "height": H, "views": len(cams), "n_visible_last": n_vis,
                   "isects_mean": M_mean, "parallelism": f"view-dp{world}",
                   "allreduce_bytes": flat.last_payload_bytes if world > 1 else 0, "allreduce_dense_bytes": flat.nbytes if world > 1 else 0},
        "roofline": roofline,
        "per_kernel_ms": {k: round(v[1], 4) for k, v in sorted(tall.items())},
        "per_kernel_ms_source": "HIP events around every operator in 6 extra steps AFTER the timed region (the timed steps bracket only the roofline kernel)",
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
