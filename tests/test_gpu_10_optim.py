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
