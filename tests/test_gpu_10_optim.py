"""-m gpu: FusedAdam (one HIP pass per tensor) against torch.optim.Adam itself -- the optimiser the reference trainer
builds (models/trainers/base.py:222) -- on the same parameters, gradients and per-group settings."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(wd):
    assert torch.cuda.is_available()
    from bilateral_driving_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 4), (1000,), (1000, 16, 3), (7,), (3, 12, 2, 4, 4)]
    base = [torch.randn(s, generator=g) for s in shapes]
    lrs = [1.6e-4, 1e-3, 5e-2, 1.25e-4, 1e-2, 2e-3]

    def make(opt_cls):
        ps = [b.clone().cuda().requires_grad_(True) for b in base]
        groups = [{"params": [p], "name": f"g{i}", "lr": lrs[i], "eps": 1e-15, "weight_decay": wd if i % 2 == 0 else 0.0} for i, p in enumerate(ps)]
        return ps, opt_cls(groups, lr=0.0, eps=1e-15)

    pa, oa = make(torch.optim.Adam)
    pb, ob = make(FusedAdam)
    for it in range(25):
        gg = torch.Generator().manual_seed(100 + it)
        for i, (x, y) in enumerate(zip(pa, pb)):
            gr = torch.randn(x.shape, generator=gg) * (10.0 ** ((i % 3) - 2))
            if it % 7 == 3 and i == 2:
                gr.zero_()                       # an all-zero gradient (culled Gaussians): m decays, eps dominates
            x.grad = gr.cuda(); y.grad = gr.cuda()
        if it == 10:                              # the reference's schedulers rewrite group["lr"] every step
            for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
                grp_a["lr"] *= 0.5; grp_b["lr"] *= 0.5
        oa.step(); ob.step()
    for i, (x, y) in enumerate(zip(pa, pb)):
        sa, sb = oa.state[x], ob.state[y]
        assert float(sa["step"]) == float(sb["step"]) == 25
        for a, b, what in ((x, y, "param"), (sa["exp_avg"], sb["exp_avg"], "exp_avg"), (sa["exp_avg_sq"], sb["exp_avg_sq"], "exp_avg_sq")):
            err = float((a.detach() - b.detach()).abs().max()) / max(float(a.detach().abs().max()), 1e-30)
            assert err < 2e-6, (i, what, err)


@pytest.mark.parametrize("consume", [False, True])
def test_fused_adam_reads_gradients_that_are_columns_of_a_row_block(consume):
    """dist.FlatGradients(row_block=True): the gradients of means / quats / log_scales / opacity logits are column ranges of one [N,16]
    block -- bds_adam_step_rows updates from (and, consuming, clears) them where they lie: the same numbers, bit for bit, as the
    contiguous pass."""
    from bilateral_driving_amd.dist import FlatGradients
    from bilateral_driving_amd.optim import FusedAdam
    N = 5003
    g = torch.Generator().manual_seed(1)
    shapes = [(N, 3), (N, 4), (N, 3), (N,), (N, 16, 3)]

    def make():
        return [torch.randn(s, generator=torch.Generator().manual_seed(7 + i)).cuda().requires_grad_(True) for i, s in enumerate(shapes)]

    pa, pb = make(), make()
    flat = FlatGradients(pb, sparse_rows=True, row_block=True)
    views = flat.arena(["means", "quats", "log_scales", "opacity_logits", "sh"])
    assert not views["quats"].is_contiguous() and views["sh"].is_contiguous() and flat.flat.numel() == N * (16 + 48)
    oa = FusedAdam([{"params": [p], "lr": 1e-3 * (i + 1)} for i, p in enumerate(pa)], lr=0.0, eps=1e-15, consume_grads=consume)
    ob = FusedAdam([{"params": [p], "lr": 1e-3 * (i + 1)} for i, p in enumerate(pb)], lr=0.0, eps=1e-15, consume_grads=consume)
    names = ["means", "quats", "log_scales", "opacity_logits", "sh"]
    for it in range(5):
        for i, (x, y) in enumerate(zip(pa, pb)):
            gr = (torch.randn(x.shape, generator=g) * 0.1).cuda()
            x.grad = gr.clone()
            views[names[i]].copy_(gr)
            y.grad = views[names[i]]
        oa.step(); ob.step()
        if consume:
            assert float(flat.flat.abs().max()) == 0.0 and all(float(x.grad.abs().max()) == 0.0 for x in pa)
        assert float(flat.flat[:N * 16].view(N, 16)[:, 11:].abs().max()) == 0.0          # the unused columns stay zero
    for x, y in zip(pa, pb):
        assert torch.equal(x.detach(), y.detach())
        assert torch.equal(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"]) and torch.equal(oa.state[x]["exp_avg"], ob.state[y]["exp_avg"])


def test_fused_adam_state_layout_allows_the_reference_surgery():
    """models/gaussians/basics.py:162-206 style: replace a parameter and its state tensors by concatenated ones."""
    from bilateral_driving_amd.optim import FusedAdam
    p = torch.randn(10, 3, device="cuda").requires_grad_(True)
    opt = FusedAdam([{"params": [p], "name": "x", "lr": 1e-2, "eps": 1e-15, "weight_decay": 0}], lr=0.0, eps=1e-15)
    p.grad = torch.randn_like(p); opt.step()
    st = opt.state.pop(p)
    new_p = torch.nn.Parameter(torch.cat([p.detach(), torch.zeros(5, 3, device="cuda")]))
    st["exp_avg"] = torch.cat([st["exp_avg"], torch.zeros(5, 3, device="cuda")])
    st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"], torch.zeros(5, 3, device="cuda")])
    opt.param_groups[0]["params"] = [new_p]
    opt.state[new_p] = st
    new_p.grad = torch.randn_like(new_p); opt.step()
    assert float(opt.state[new_p]["step"]) == 2 and torch.isfinite(new_p).all()


def test_densify_stats_match_the_reference_method(golden_dir):
    """optim.DensifyStats vs tests/golden/densify_stats.npz = three consecutive calls of the reference's own
    VanillaGaussians.after_train fed as BasicTrainer.postprocess_per_train_step feeds it."""
    import os
    import numpy as np
    from bilateral_driving_amd.optim import DensifyStats
    z = np.load(os.path.join(golden_dir, "densify_stats.npz"))
    N, W, H = int(z["N"]), int(z["W"]), int(z["H"])
    st = DensifyStats(N, "cuda", batch_size=int(z["batch"]))
    for call in range(3):
        m2 = torch.zeros(1, N, 2, device="cuda")
        m2.absgrad = torch.from_numpy(z[f"absgrad{call}"]).cuda()
        info = {"means2d": m2, "radii": torch.from_numpy(z[f"radii{call}"]).cuda()[None], "width": W, "height": H}
        st.update(info)
        for name in ("xys_grad_norm", "vis_counts", "max_2Dsize"):
            ref = torch.from_numpy(z[f"{name}{call}"]).cuda()
            got = getattr(st, name)
            assert torch.allclose(got, ref, rtol=2e-7, atol=0), (call, name, float((got - ref).abs().max()))


@pytest.mark.parametrize("wd,table_steps", [(0.0, 1024), (0.0, 16), (0.01, 16)])
def test_deferred_row_adam_is_bit_equal_to_the_dense_pass(wd, table_steps):
    """optim.DeferredRowAdam: the SH rows step only when a view's list names them, the missed zero-gradient steps are replayed when the
    row is next read (catchup) -- parameters and both moments equal the dense FusedAdam (= torch.optim.Adam, first test) BIT FOR BIT
    after K steps of random visibility, a per-step lr schedule, rows listed by two views of a step, device-side counts, a ring table
    that wraps (table_steps = 16: the automatic flush) and weight decay (a zero gradient is then not a zero update).  The reference's
    cadence: one dense Adam step per single-view iteration (/root/reference/project/models/trainers/base.py:222-226,502-516)."""
    from bilateral_driving_amd.optim import DeferredRowAdam, FusedAdam
    N, K, steps = 5003, 16, 60
    gen = torch.Generator().manual_seed(3)
    sh0 = torch.randn(N, K, 3, generator=gen).cuda()
    mu0 = torch.randn(N, 3, generator=gen).cuda()
    # dense reference: dc and rest as two tensors with their own rates (the reference's _features_dc / _features_rest)
    dc = sh0[:, :1].clone().contiguous().requires_grad_(True)
    rest = sh0[:, 1:].clone().contiguous().requires_grad_(True)
    mu_a = mu0.clone().requires_grad_(True)
    oa = FusedAdam([{"params": [dc], "lr": 2.5e-3, "weight_decay": wd}, {"params": [rest], "lr": 2.5e-3 / 20, "weight_decay": wd},
                    {"params": [mu_a], "lr": 1.6e-4}], lr=0.0, eps=1e-15)
    sh = sh0.clone().requires_grad_(True)
    mu_b = mu0.clone().requires_grad_(True)
    ob = DeferredRowAdam([{"params": [sh], "lr": 2.5e-3, "lr_b": 2.5e-3 / 20, "col_split": 3, "deferred_rows": True, "weight_decay": wd},
                          {"params": [mu_b], "lr": 1.6e-4}], lr=0.0, eps=1e-15, table_steps=table_steps)
    sh.grad = torch.zeros_like(sh)
    cap = 1500

    def make_list():
        n = int(torch.randint(200, 1200, (1,), generator=gen))
        ids = torch.randperm(N, generator=gen)[:n].sort().values.to(torch.int32)
        pad = torch.full((cap,), -1, dtype=torch.int32)
        pad[:n] = ids
        return pad.cuda(), torch.tensor([n], dtype=torch.int64).cuda(), ids.long().cuda()

    for it in range(steps):
        lists = [make_list() for _ in range(1 if it % 3 else 2)]
        if it == 20:
            lists = []                                    # a step without any view
        # the view's forward: the listed rows are brought up to date before they are read
        for ids, cnt, rows in lists:
            ob.catchup(cap, cnt.data_ptr(), ids)
            assert torch.equal(sh.detach()[rows, :1], dc.detach()[rows]) and torch.equal(sh.detach()[rows, 1:], rest.detach()[rows])
        # the backward: gradient rows of the listed Gaussians only (dense zeros elsewhere, as gsplat's dense gradients)
        g = torch.zeros(N, K, 3, device="cuda")
        for ids, cnt, rows in lists:
            g[rows] += torch.randn(rows.numel(), K, 3, generator=gen).cuda() * 0.01
        if it % 7 == 5 and lists:
            g[lists[0][2][:50]] = 0.0                     # visible rows whose gradient is exactly zero
        dc.grad, rest.grad = g[:, :1].contiguous(), g[:, 1:].contiguous()
        sh.grad.copy_(g)
        gm = torch.randn(N, 3, generator=gen).cuda()
        mu_a.grad, mu_b.grad = gm.clone(), gm.clone()
        for grp_a in oa.param_groups:                     # the reference's schedulers rewrite group["lr"] every step
            grp_a["lr"] *= 0.99
        ob.param_groups[0]["lr"] *= 0.99; ob.param_groups[0]["lr_b"] *= 0.99; ob.param_groups[1]["lr"] *= 0.99
        oa.step()
        ob.step(lists=[(cap, cnt.data_ptr(), ids) for ids, cnt, rows in lists])
    assert torch.equal(mu_a.detach(), mu_b.detach())
    stale = int((ob.state[sh]["last_step"] < steps).sum())
    assert table_steps == 16 or stale > 0                 # (rows no list has named since: still behind)
    ob.flush()
    assert int((ob.state[sh]["last_step"] != steps).sum()) == 0 and float(ob.state[sh]["step"]) == steps
    for got, a, b in ((sh.detach(), dc.detach(), rest.detach()),
                      (ob.state[sh]["exp_avg"], oa.state[dc]["exp_avg"], oa.state[rest]["exp_avg"]),
                      (ob.state[sh]["exp_avg_sq"], oa.state[dc]["exp_avg_sq"], oa.state[rest]["exp_avg_sq"])):
        assert torch.equal(got[:, :1], a) and torch.equal(got[:, 1:], b)
    # every row at once (lists=None) is the dense pass through the same kernel
    g = torch.randn(N, K, 3, generator=gen).cuda()
    dc.grad, rest.grad = g[:, :1].contiguous(), g[:, 1:].contiguous()
    sh.grad.copy_(g)
    mu_a.grad, mu_b.grad = None, None
    oa.step(); ob.step()
    assert torch.equal(sh.detach()[:, :1], dc.detach()) and torch.equal(sh.detach()[:, 1:], rest.detach())


def test_deferred_row_adam_checkpoint_round_trip_continues_bit_equal():
    """state_dict() flushes (every row at the step of the save); a fresh optimizer that loads it continues exactly where the first one
    would have: parameters and moments bit-equal to the uninterrupted run after more lazy steps (the reference resumes model-only,
    trainers/base.py:683-705 -- an optimizer checkpoint is this package's addition and must not change the numbers)."""
    from bilateral_driving_amd.optim import DeferredRowAdam
    N, K = 3001, 16
    gen = torch.Generator().manual_seed(11)
    sh0 = torch.randn(N, K, 3, generator=gen).cuda()

    def make(x):
        p = x.clone().requires_grad_(True)
        p.grad = torch.zeros_like(p)
        return p, DeferredRowAdam([{"params": [p], "lr": 2.5e-3, "lr_b": 1.25e-4, "col_split": 3, "deferred_rows": True}], lr=0.0, eps=1e-15)

    def steps(p, opt, k0, k1):
        for it in range(k0, k1):
            g = torch.Generator().manual_seed(500 + it)
            ids = torch.randperm(N, generator=g)[:400].sort().values
            lst = ids.to(torch.int32).cuda()
            opt.catchup(lst.numel(), None, lst)
            p.grad.zero_()
            p.grad[ids.cuda()] = (torch.randn(400, K, 3, generator=g) * 0.01).cuda()
            opt.step(lists=[(lst.numel(), None, lst)])

    pa, oa = make(sh0)
    steps(pa, oa, 0, 30)                      # the uninterrupted run
    pb, ob = make(sh0)
    steps(pb, ob, 0, 17)
    sd = ob.state_dict()
    assert int((ob.state[pb]["last_step"] != 17).sum()) == 0
    pc, oc = make(pb.detach())                # "resume": parameters from the model checkpoint, optimizer from its own
    oc.load_state_dict(sd)
    assert oc._t == 17 and oc.state[pc]["last_step"].dtype == torch.int32 and int(oc._clock[0]) == 17
    steps(pc, oc, 17, 30)
    oa.flush(); oc.flush()
    assert torch.equal(pa.detach(), pc.detach())
    assert torch.equal(oa.state[pa]["exp_avg"], oc.state[pc]["exp_avg"]) and torch.equal(oa.state[pa]["exp_avg_sq"], oc.state[pc]["exp_avg_sq"])
