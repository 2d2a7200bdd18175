"""world_size-2 gloo tests (CPU) of the multi-GPU exchange step: flat gradient buffer + one
sum-all-reduce == sequential accumulation over the same views on one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_loss(params, view):
    # any differentiable function of the shared parameters that depends on the view
    w = torch.linspace(0.5, 1.5, params[0].numel()).reshape(params[0].shape) * (view + 1)
    return (params[0] * w).sum() + (params[1] ** 2).sum() * (view + 2) + (params[2].sin() * (view + 1)).sum()


def _make_params():
    g = torch.Generator().manual_seed(0)
    return [torch.randn(7, 3, generator=g).requires_grad_(True), torch.randn(7, 16, 3, generator=g).requires_grad_(True),
            torch.randn(5, generator=g).requires_grad_(True)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bilateral_driving_amd.dist import FlatGradients, reduce_densify_stats, view_for_rank
    params = _make_params()
    flat = FlatGradients(params)
    outs = []
    for step in range(3):
        flat.zero()
        v = view_for_rank(step, rank, world, 6)
        _toy_loss(params, v).backward()
        flat.all_reduce()
        assert all(p.grad.data_ptr() == w.data_ptr() for p, w in zip(params, flat._views))  # grads = reduced slices
        outs.append(flat.flat.clone())
    a, b, c = torch.full((4,), float(rank + 1)), torch.full((4,), float(rank + 1)), torch.tensor([1.0, 5.0, 2.0, 0.0]) * (rank + 1)
    reduce_densify_stats(a, b, c)
    q.put((rank, [o.numpy() for o in outs], a.numpy(), c.numpy()))
    dist.destroy_process_group()


def test_flat_allreduce_equals_sequential_sum():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    from bilateral_driving_amd.dist import view_for_rank
    for step in range(3):
        params = _make_params()
        for r in range(world):
            _toy_loss(params, view_for_rank(step, r, world, 6)).backward()  # sequential accumulation on one process
        ref = torch.cat([p.grad.reshape(-1) for p in params]).numpy()
        for r in range(world):
            assert abs(res[r][1][step] - ref).max() < 1e-5
    assert res[0][2].tolist() == [3.0] * 4           # sum of (1, 2)
    assert res[0][3].tolist() == [2.0, 10.0, 4.0, 0.0]  # max


def test_flat_gradients_single_process():
    from bilateral_driving_amd.dist import FlatGradients
    params = _make_params()
    flat = FlatGradients(params)
    assert flat.nbytes == 4 * sum(p.numel() for p in params)
    _toy_loss(params, 0).backward()
    g0 = torch.cat([p.grad.reshape(-1) for p in params])
    assert flat.all_reduce() is None  # no process group: no-op, nothing is packed or copied
    assert flat._flat is None
    assert torch.equal(flat.pack(), g0)
    flat.zero()
    assert all(p.grad is None for p in params)
    _toy_loss(params, 0).backward()
    assert torch.equal(flat.pack(), g0)


def _rows_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bilateral_driving_amd.dist import FlatGradients
    N = 200
    g = torch.Generator().manual_seed(1)
    params = [torch.randn(N, 3, generator=g).requires_grad_(True), torch.randn(N, 16, 3, generator=g).requires_grad_(True),
              torch.randn(N, generator=g).requires_grad_(True), torch.randn(6, 12, 2, 2, 2, generator=g).requires_grad_(True)]
    flat = FlatGradients(params)
    outs, payloads = [], []
    for step in range(3):
        flat.zero()
        vis = torch.zeros(N, dtype=torch.bool)
        lo = (37 * step + 60 * rank) % N
        vis[lo:lo + 30] = True                                     # each rank touches its own 15 % of the rows
        w = vis.float()
        loss = (params[0] * w[:, None] * (rank + 1)).sum() + (params[1] ** 2 * w[:, None, None]).sum() + (params[2].sin() * w).sum() \
            + (params[3] ** 2).sum() * (rank + 2)                  # the last one is dense on every rank (grids / TV)
        flat.begin_rows_union(vis)
        loss.backward()
        flat.all_reduce()
        outs.append(flat.flat.clone())
        payloads.append(flat.last_payload_bytes)
    q.put((rank, [o.numpy() for o in outs], payloads, flat.nbytes))
    dist.destroy_process_group()


def test_rows_union_exchange_equals_dense_sum():
    """begin_rows_union + all_reduce (only rows touched on some rank travel) == dense all-reduce, with a smaller payload."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    N = 200
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(N, 3, generator=g), torch.randn(N, 16, 3, generator=g), torch.randn(N, generator=g), torch.randn(6, 12, 2, 2, 2, generator=g)]
    for step in range(3):
        params = [b.clone().requires_grad_(True) for b in base]
        for rank in range(world):
            vis = torch.zeros(N)
            lo = (37 * step + 60 * rank) % N
            vis[lo:lo + 30] = 1.0
            ((params[0] * vis[:, None] * (rank + 1)).sum() + (params[1] ** 2 * vis[:, None, None]).sum() + (params[2].sin() * vis).sum()
             + (params[3] ** 2).sum() * (rank + 2)).backward()
        ref = torch.cat([p.grad.reshape(-1) for p in params]).numpy()
        for r in range(world):
            assert abs(res[r][1][step] - ref).max() < 1e-5
            assert res[r][2][step] < 0.5 * res[r][3]               # the payload really shrank (<= 60 of 200 rows + the dense tail)


def _randn_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bilateral_driving_amd.dist import broadcast_randn
    torch.manual_seed(100 + rank)                      # different streams: the broadcast must make the ranks agree
    t = broadcast_randn((7, 3), "cpu")
    e = broadcast_randn((0, 3), "cpu")
    q.put((rank, t.numpy(), tuple(e.shape)))
    dist.destroy_process_group()


def test_split_noise_is_rank0s_draw_on_every_rank():
    """Densification under view parallelism: every replica has to split with the same noise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_randn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])
    torch.manual_seed(100)
    np.testing.assert_array_equal(res[0][1], torch.randn(7, 3).numpy())
    assert res[0][2] == (0, 3) and res[1][2] == (0, 3)
    from bilateral_driving_amd.dist import broadcast_randn      # no process group: plain randn
    assert broadcast_randn((2, 3), "cpu").shape == (2, 3)


def test_sparse_rows_bookkeeping_single_process():
    """FlatGradients(sparse_rows=True) on CPU tensors (the row-wise clear itself is a GPU kernel; here its torch equivalent runs):
    the state machine that decides when a producer may rely on an all-zero buffer."""
    from bilateral_driving_amd.dist import FlatGradients
    params = _make_params()
    flat = FlatGradients(params, sparse_rows=True)
    flat.arena(["means", "sh", "tail"])
    assert not flat.rows_clean                        # unknown until the first zero()
    flat.mark_rows(torch.tensor([True] * 7))          # ignored while the state is unknown
    assert flat._dirty is None
    flat.zero()
    assert flat.rows_clean and float(flat.flat.abs().sum()) == 0.0
    flat.mark_list(torch.tensor([0, 3], dtype=torch.int32))
    assert not flat.rows_clean
    flat.mark_rows(torch.tensor([0, 0, 1, 1, 0, 0, 0], dtype=torch.bool))   # the views of a frame: the lists accumulate
    assert sorted(set(int(i) for ids in flat._dirty for i in ids.tolist())) == [0, 2, 3]
    n_row_floats = 7 * 3 + 7 * 16 * 3
    rows = flat.flat[:n_row_floats]
    flat._views[0][[0, 2, 3]] = 3.0                   # what the backward of those views wrote
    flat._views[1][[0, 2, 3]] = 3.0
    flat.zero()
    assert flat.rows_clean and flat._dirty == [] and float(rows.abs().sum()) == 0.0
    # an exchange without a union (dense all-reduce) makes the book unknown again: covered in the two-process test below
    dense = FlatGradients(_make_params())
    dense.zero()
    assert not dense.rows_clean                       # the default form never promises anything


# ---- frame-wise exchange (dist.FrameExchange): per-view compact all-reduce, logic on CPU tensors over gloo ------------------
_FX_N, _FX_K, _FX_VIEWS, _FX_FRAMES = 300, 4, 3, 3
_FX_NAMES = ["means", "log_scales", "quats", "opacity_logits", "sh", "grid0"]


def _fx_params():
    g = torch.Generator().manual_seed(5)
    shapes = [(_FX_N, 3), (_FX_N, 3), (_FX_N, 4), (_FX_N,), (_FX_N, _FX_K, 3), (2, 12, 1, 2, 2)]
    return [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]


def _fx_view(rank, frame, v):
    """What a fused view's backward would produce on `rank`: the visible id list and one gradient row per visible Gaussian."""
    g = torch.Generator().manual_seed(1000 * rank + 10 * frame + v)
    lo = (41 * frame + 70 * v + 55 * rank) % _FX_N
    n = 40 + 5 * v + (30 if frame == 2 else 0)               # the last frame's unions are larger: the capacity has to hold
    ids = (torch.arange(lo, lo + n) % _FX_N).sort().values.to(torch.int32)
    rows = {"means": torch.randn(n, 3, generator=g), "log_scales": torch.randn(n, 3, generator=g), "quats": torch.randn(n, 4, generator=g),
            "opacity_logits": torch.randn(n, generator=g), "sh": torch.randn(n, _FX_K, 3, generator=g)}
    grid_grad = torch.randn(2, 12, 1, 2, 2, generator=g)
    return ids, rows, grid_grad


def _fx_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    params = _fx_params()
    flat = FlatGradients(params, sparse_rows=True)
    fx = FrameExchange(flat, _FX_NAMES, headroom=1.6)
    outs, payloads = [], []
    for frame in range(_FX_FRAMES):
        fx.begin_frame()
        for v in range(_FX_VIEWS):
            assert fx.view_kwargs(v) == dict(grad_sink=fx)
            ids, rows, grid_grad = _fx_view(rank, frame, v)
            radii = torch.zeros(1, _FX_N, dtype=torch.int32)
            radii[0, ids.long()] = 3
            fx.begin_view({"radii": radii, "visible_ids": ids})
            # ... what fused_view's backward does with the sink: rows of the visible Gaussians at their union slots
            bufs, row_map = fx.targets(ids)
            slots = row_map[ids.long()].long()
            for k, r in rows.items():
                bufs[k][slots] = r
            params[5].grad = grid_grad.clone() if params[5].grad is None else params[5].grad + grid_grad   # autograd's accumulation
            fx.end_view()
        fx.end_frame()
        assert all(p.grad is not None and p.grad.data_ptr() == w.data_ptr() for p, w in zip(params, flat._views))
        outs.append(flat.flat.clone())
        payloads.append(fx.payload_bytes)
    q.put((rank, [o.numpy() for o in outs], payloads, flat.nbytes, fx.cap))
    dist.destroy_process_group()


def test_frame_exchange_equals_sequential_sum_over_views_and_ranks():
    """Every frame: sum over ranks and views of the per-view gradient rows, exchanged per view through compact buffers whose
    slots are the union of the ranks' visible sets == one process adding up the same rows; every single exchange moves a fraction
    of the dense buffer (what makes it short enough to hide behind the next view); stale rows of the previous frame are cleared
    through the union lists."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fx_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = [tuple(p.shape) for p in _fx_params()]
    for frame in range(_FX_FRAMES):
        ref = [torch.zeros(s) for s in shapes]
        for rank in range(world):
            for v in range(_FX_VIEWS):
                ids, rows, grid_grad = _fx_view(rank, frame, v)
                for i, k in enumerate(_FX_NAMES[:5]):
                    ref[i].index_add_(0, ids.long(), rows[k])
                ref[5] += grid_grad
        ref = torch.cat([r.reshape(-1) for r in ref]).numpy()
        for r in range(world):
            assert abs(res[r][1][frame] - ref).max() < 1e-5, (frame, r)
            assert res[r][2][frame] / _FX_VIEWS < 0.8 * res[r][3]       # every single exchange moves a fraction of the dense buffer
    assert res[0][4] == res[1][4] and res[0][4] % 4 == 0               # same capacity on every rank, 16-byte aligned sub-arrays


def _fx_frame_worker(rank, world, port, q):
    """FrameExchange(per_view=False): every view accumulates in place as at world size 1 (arena modes); ONE dense all-reduce per frame."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    params = _fx_params()
    flat = FlatGradients(params, sparse_rows=True)
    fx = FrameExchange(flat, _FX_NAMES, per_view=False)
    assert not fx.active and fx.frame_reduce and fx.world == world
    outs = []
    for frame in range(_FX_FRAMES):
        fx.begin_frame()
        tail = fx.tail_grads()
        for v in range(_FX_VIEWS):
            kw = fx.view_kwargs(v)
            assert kw["arena_rows"] == (1 if v == 0 else 2)
            ids, rows, grid_grad = _fx_view(rank, frame, v)
            fx.begin_view({"radii": None, "visible_ids": ids})
            for k, r in rows.items():     # ... the fused view's backward, arena mode: rows of the visible Gaussians added in place
                kw["grad_arena"][k].index_add_(0, ids.long(), r)
            tail[0].add_(grid_grad)
            fx.end_view()
        fx.end_frame()
        outs.append(flat.flat.clone())
        assert fx.payload_bytes == flat.nbytes and fx.n_exchanges == 1
    q.put((rank, [o.numpy() for o in outs]))
    dist.destroy_process_group()


def test_frame_exchange_per_frame_mode_equals_sequential_sum():
    """The other exchange: one dense all-reduce at the end of the frame; rows the OTHER ranks wrote are cleared before the next frame."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fx_frame_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = [tuple(p.shape) for p in _fx_params()]
    for frame in range(_FX_FRAMES):
        ref = [torch.zeros(s) for s in shapes]
        for rank in range(world):
            for v in range(_FX_VIEWS):
                ids, rows, grid_grad = _fx_view(rank, frame, v)
                for i, k in enumerate(_FX_NAMES[:5]):
                    ref[i].index_add_(0, ids.long(), rows[k])
                ref[5] += grid_grad
        ref = torch.cat([r.reshape(-1) for r in ref]).numpy()
        for r in range(world):
            assert abs(res[r][1][frame] - ref).max() < 1e-5, (frame, r)


def test_plan_exchange_prices_the_two_exchanges():
    """dist.plan_exchange on the sizes measured for the benchmark scene (profiles/r05i_bench_share*.json: unions of 18 / 27 / 47 % of 2 M
    Gaussians at 2 / 4 / 8 ranks, 5.9 ms of compute per frame): per view while the exchanges hide behind the next view, per frame once
    the wire time of six union buffers exceeds the frame; a slow fabric flips the choice earlier, a fast one later."""
    from bilateral_driving_amd.dist import plan_exchange
    N = 2_000_000
    plans = {w: plan_exchange([int(f * N)] * 6, N, 59, 50_000, w, 5.9e-3, 300e9) for w, f in ((2, 0.18), (4, 0.27), (8, 0.47))}
    assert plans[2]["per_view"] and plans[4]["per_view"] and not plans[8]["per_view"]
    assert plans[8]["per_frame_bytes"] == (N * 59 + 50_000) * 4 and plans[8]["per_view_bytes"] > 3 * plans[8]["per_frame_bytes"]
    assert abs(plans[8]["exposed_per_frame_ms"] - 1.75 * (N * 59 + 50_000) * 4 / 300e9 * 1e3) < 1e-6
    assert not plan_exchange([int(0.27 * N)] * 6, N, 59, 50_000, 4, 5.9e-3, 100e9)["per_view"]      # a third of the bandwidth: nothing hides
    assert plan_exchange([int(0.47 * N)] * 6, N, 59, 50_000, 8, 5.9e-3, 900e9)["per_view"]
    one = plan_exchange([N // 10], N, 59, 0, 1, 1e-3, 0.0)                                            # world size 1: nothing on the wire
    assert one["exposed_per_view_ms"] == 0.0 and one["exposed_per_frame_ms"] == 0.0


def test_frame_exchange_single_process_uses_the_arena_modes():
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    params = _fx_params()
    flat = FlatGradients(params, sparse_rows=True)
    fx = FrameExchange(flat, _FX_NAMES)
    assert not fx.active
    fx.begin_frame()
    kw0, kw1 = fx.view_kwargs(0), fx.view_kwargs(1)
    assert kw0["arena_rows"] == 1 and kw1["arena_rows"] == 2 and kw0["grad_arena"]["sh"].data_ptr() == flat._views[4].data_ptr()
    fx.begin_view({"radii": torch.ones(1, _FX_N, dtype=torch.int32), "visible_ids": torch.arange(5, dtype=torch.int32)})
    fx.end_view()
    fx.end_frame()
    assert flat._dirty is not None and len(flat._dirty) == 1
    # the small dense tail (grids) is accumulated in place at world size 1: begin_frame zeroes it and makes it the .grad
    tail = fx.tail_grads()
    assert tail is not None and len(tail) == 1 and tail[0].shape == params[5].shape
    assert params[5].grad is not None and params[5].grad.data_ptr() == tail[0].data_ptr() and float(tail[0].abs().max()) == 0.0
    tail[0].add_(1.0)
    fx.begin_frame()
    assert float(fx.tail_grads()[0].abs().max()) == 0.0 and params[5].grad.data_ptr() == fx.tail_grads()[0].data_ptr()
    assert fx.tail_grads("nope") == []


def test_frame_exchange_capacity_follows_the_largest_view_and_overflow_raises_before_the_step():
    """Views of very different visibility (advisor finding, round 2): the exchange capacity is sized from the LARGEST union of the
    first frame, not from the first view; a later union that outgrows it raises at end_frame -- before an optimizer step could consume
    the incomplete gradients -- and the next frame runs with re-sized buffers.  Compact path without collectives (force=True)."""
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    params = _fx_params()
    flat = FlatGradients(params, sparse_rows=True)
    fx = FrameExchange(flat, _FX_NAMES, headroom=1.25, force=True)

    def run_frame(sizes, seed, fits=True):
        g = torch.Generator().manual_seed(seed)
        ref = [torch.zeros_like(p) for p in params[:5]]
        fx.begin_frame()
        for v, n in enumerate(sizes):
            ids = torch.randperm(_FX_N, generator=g)[:n].sort().values.to(torch.int32)
            radii = torch.zeros(1, _FX_N, dtype=torch.int32)
            radii[0, ids.long()] = 2
            fx.begin_view({"radii": radii, "visible_ids": ids})
            bufs, row_map = fx.targets(ids)
            slots = row_map[ids.long()].long()
            assert not fits or slots.unique().numel() == n, "two Gaussians share an exchange slot"
            for i, k in enumerate(_FX_NAMES[:5]):
                r = torch.randn(n, *params[i].shape[1:], generator=g)
                bufs[k][slots] = r
                ref[i].index_add_(0, ids.long(), r)
            fx.end_view()
        fx.end_frame()
        for i in range(5):
            assert torch.allclose(flat._views[i], ref[i], atol=1e-6), i

    run_frame([20, 150, 60], 1)             # view 1 sees 7.5x what view 0 sees
    cap0 = fx.cap
    assert cap0 >= int(150 * 1.25) and cap0 % 4 == 0
    run_frame([30, 140, 150], 2)            # fits: nothing changes
    assert fx.cap == cap0
    with pytest.raises(RuntimeError, match="outgrew the exchange capacity"):
        run_frame([30, cap0 + 40, 50], 3, fits=False)   # (slots beyond the capacity collide) raised by end_frame, i.e. before the caller's optimizer step
    run_frame([30, cap0 + 40, 50], 3)       # the repeated frame runs with re-sized buffers
    assert fx.cap >= cap0 + 40 and fx.cap >= cap0
    run_frame([10, 20, 30], 4)
    assert fx.cap >= cap0 + 40              # never shrinks


# ---- the many-rank regime: the union of the ranks' visible sets approaches the whole scene ----------------------------------------
def _fx_wide_view(rank, world, frame, v):
    """Every rank sees its own contiguous ~1/world of the Gaussians (+ an overlap), so that the UNION of a view over the ranks covers
    >= 90 % of the scene while each rank still writes a small fraction of the rows -- what 8 timesteps of a drive look like."""
    g = torch.Generator().manual_seed(7000 * rank + 10 * frame + v)
    per = _FX_N // world
    lo = rank * per + 3 * v
    n = min(per + 12, _FX_N - 1)
    ids = (torch.arange(lo, lo + n) % _FX_N).sort().values.to(torch.int32)
    rows = {"means": torch.randn(n, 3, generator=g), "log_scales": torch.randn(n, 3, generator=g), "quats": torch.randn(n, 4, generator=g),
            "opacity_logits": torch.randn(n, generator=g), "sh": torch.randn(n, _FX_K, 3, generator=g)}
    return ids, rows


def _fx_wide_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    params = _fx_params()
    flat = FlatGradients(params, sparse_rows=True)
    fx = FrameExchange(flat, _FX_NAMES, headroom=1.25)
    outs, caps, payloads, unions = [], [], [], []
    for frame in range(2):
        fx.begin_frame()
        for v in range(_FX_VIEWS):
            ids, rows = _fx_wide_view(rank, world, frame, v)
            radii = torch.zeros(1, _FX_N, dtype=torch.int32)
            radii[0, ids.long()] = 3
            fx.begin_view({"radii": radii, "visible_ids": ids})
            bufs, row_map = fx.targets(ids)
            slots = row_map[ids.long()].long()
            assert slots.unique().numel() == ids.numel()
            for k, r in rows.items():
                bufs[k][slots] = r
            fx.end_view()
        fx.end_frame()
        outs.append(flat.flat.clone())
        caps.append(fx.cap)
        payloads.append(fx.payload_bytes)
        unions.append(fx._max_union)
    q.put((rank, [o.numpy() for o in outs], caps, payloads, unions, fx.row_floats))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_frame_exchange_when_the_union_approaches_the_whole_scene(world):
    """4 / 8 ranks, every view's union >= 90 % of the Gaussians (the regime DESIGN.md section 6 prices: the compact exchange then
    moves about what a dense all-reduce moves): the capacity grows to the union in the first frame and is capped at the scene
    (rounded up to 4 rows), nothing overflows, the payload is views x capacity x row bytes, the sums are right on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fx_wide_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = [tuple(p.shape) for p in _fx_params()]
    n_row = sum(int(torch.tensor(s).prod()) for s in shapes[:5])
    for frame in range(2):
        ref = [torch.zeros(s) for s in shapes[:5]]
        for rank in range(world):
            for v in range(_FX_VIEWS):
                ids, rows = _fx_wide_view(rank, world, frame, v)
                for i, k in enumerate(_FX_NAMES[:5]):
                    ref[i].index_add_(0, ids.long(), rows[k])
        ref = torch.cat([r.reshape(-1) for r in ref]).numpy()
        for r in range(world):
            assert abs(res[r][1][frame][:n_row] - ref).max() < 1e-5, (frame, r)
    cap_max = (_FX_N + 3) // 4 * 4
    for r in range(world):
        rank, _, caps, payloads, unions, row_floats = res[r]
        assert unions[-1] >= 0.9 * _FX_N, unions                       # the regime this test is about
        assert caps[0] == caps[1] == res[0][2][0] and unions[-1] <= caps[-1] <= cap_max, (caps, unions)
        assert payloads[1] >= _FX_VIEWS * caps[1] * row_floats * 4      # (+ the dense tail)


def _bound_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bilateral_driving_amd.dist import dynamic_union_bound
    q.put((rank, dynamic_union_bound(100 * (rank + 1), 10_000), dynamic_union_bound(4000 + rank, 10_000)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_capacity_protocol_of_replayable_view_slots_with_an_exchange(world):
    """graph_view.FrameGraph(dynamic=True, exchange=..): every rank draws its own camera per step, so a slot's union over the ranks is
    bounded by the SUM of the ranks' largest visible sets, capped at the scene -- one collective, the same number on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bound_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sum(100 * (r + 1) for r in range(world))
    assert all(r[1] == want for r in res)
    assert all(r[2] == min(10_000, sum(4000 + r for r in range(world))) for r in res)       # (world 3: capped at the scene)
    from bilateral_driving_amd.dist import dynamic_union_bound          # no process group: the rank's own maximum
    assert dynamic_union_bound(123, 1000) == 123 and dynamic_union_bound(5000, 1000) == 1000


def _forced_world1_worker(port, q):
    """ONE rank, gloo, ``force_collectives``: the per-view and the per-frame exchange issue every collective and give the plain sum."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from bilateral_driving_amd import dist as D
    assert not D._active()
    D.force_collectives(True)
    assert D._active() and D.collectives_forced()
    res = {}
    for per_view in (True, False):
        params = _fx_params()
        flat = D.FlatGradients(params, sparse_rows=True)
        fx = D.FrameExchange(flat, _FX_NAMES, headroom=1.6, per_view=per_view)
        assert fx.coll and fx.world == 1 and fx.active == per_view and fx.frame_reduce == (not per_view)
        n0 = D.ISSUED["all_reduce"]
        outs = []
        for frame in range(_FX_FRAMES):
            fx.begin_frame()
            for v in range(_FX_VIEWS):
                ids, rows, grid_grad = _fx_view(0, frame, v)
                radii = torch.zeros(1, _FX_N, dtype=torch.int32)
                radii[0, ids.long()] = 3
                fx.begin_view({"radii": radii, "visible_ids": ids})
                if per_view:
                    bufs, row_map = fx.targets(ids)
                    slots = row_map[ids.long()].long()
                    for k, r in rows.items():
                        bufs[k][slots] = r
                else:       # the frame accumulates in place, as at world size 1
                    for k, r in rows.items():
                        fx.arena[k].index_add_(0, ids.long(), r)
                    for p_, v_ in zip(flat.params[:5], flat._views[:5]):
                        p_.grad = v_
                params[5].grad = grid_grad.clone() if params[5].grad is None else params[5].grad + grid_grad
                fx.end_view()
            if not per_view:
                flat.pack()
            fx.end_frame()
            outs.append(flat.flat.clone().numpy())
        res[per_view] = (outs, D.ISSUED["all_reduce"] - n0)
    D.force_collectives(None)
    assert not D._active()
    q.put(res)
    dist.destroy_process_group()


def test_forced_collectives_at_world_size_1_issue_every_collective_and_change_nothing():
    """dist.force_collectives (BDS_FORCE_COLLECTIVES=1): what lets ONE GPU execute the RCCL call sequence of the multi-GPU exchange
    (tests/test_gpu_24_rccl_world1.py); here the same switch over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_world1_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0
    shapes = [tuple(t.shape) for t in _fx_params()]
    for per_view in (True, False):
        outs, issued = res[per_view]
        # per view: mask MAX + rows SUM per view + the tail per frame; per frame: ONE dense all-reduce
        assert issued == (_FX_FRAMES * (2 * _FX_VIEWS + 1) if per_view else _FX_FRAMES), (per_view, issued)
        for frame in range(_FX_FRAMES):
            ref = [torch.zeros(s) for s in shapes]
            for v in range(_FX_VIEWS):
                ids, rows, grid_grad = _fx_view(0, frame, v)
                for i, k in enumerate(_FX_NAMES[:5]):
                    ref[i].index_add_(0, ids.long(), rows[k])
                ref[5] += grid_grad
            ref = torch.cat([r.reshape(-1) for r in ref]).numpy()
            assert abs(outs[frame] - ref).max() < 1e-5, (per_view, frame)
