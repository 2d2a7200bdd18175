"""Adaptive density control on the device (bilateral_driving_amd.densify.refinement_after -> csrc/refine.hip through the C ABI)
against (1) the golden vectors of the reference's own VanillaGaussians.refinement_after and (2) the pinned CPU oracle at a size
where the plan spans thousands of workgroups."""
import glob
import math
import os
import types

import numpy as np
import pytest
import torch

from oracle import refine_oracle as RO

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refine_step*.npz")))
GROUPS = ("xyz", "sh_dc", "sh_rest", "opacity", "scaling", "rotation")


class Cfg(dict):
    __getattr__ = dict.__getitem__


def build_model(P, M, V, stats, ctrl, scene_scale, num_train_images, step, opt_cls, dev="cuda"):
    model = types.SimpleNamespace(ctrl_cfg=Cfg(ctrl), scene_scale=scene_scale, num_train_images=num_train_images, step=step,
                                  class_prefix="Background#")
    groups = []
    for a, n in zip(RO.PARAMS, GROUPS):
        prm = torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(P[a])).to(dev))
        setattr(model, a, prm)
        groups.append({"params": [prm], "lr": 1e-3, "eps": 1e-15, "weight_decay": 0, "name": model.class_prefix + n})
    opt = opt_cls(groups, lr=0.0, eps=1e-15)
    for a in RO.PARAMS:
        opt.state[getattr(model, a)] = {"step": torch.tensor(1.0), "exp_avg": torch.from_numpy(np.ascontiguousarray(M[a])).to(dev),
                                        "exp_avg_sq": torch.from_numpy(np.ascontiguousarray(V[a])).to(dev)}
    for k, v in stats.items():
        setattr(model, k, None if v is None else torch.from_numpy(v).to(dev))
    return model, opt


def check(model, opt, exp_P, exp_M, exp_V, means_atol=3e-6):
    for a, n in zip(RO.PARAMS, GROUPS):
        prm = getattr(model, a)
        assert isinstance(prm, torch.nn.Parameter) and prm.is_cuda
        got = prm.detach().cpu().numpy()
        assert got.shape == exp_P[a].shape, a
        if a in ("_means", "_scales", "_opacities"):     # exp / log / rotation: last bits
            np.testing.assert_allclose(got, exp_P[a], rtol=3e-6, atol=means_atol if a == "_means" else 3e-6, err_msg=a)
        else:
            np.testing.assert_array_equal(got, exp_P[a], err_msg=a)
        grp = [g for g in opt.param_groups if g["name"] == model.class_prefix + n][0]
        assert grp["params"][0] is prm and len(grp["params"]) == 1
        st = opt.state[prm]
        np.testing.assert_array_equal(st["exp_avg"].cpu().numpy(), exp_M[a], err_msg="exp_avg" + a)
        np.testing.assert_array_equal(st["exp_avg_sq"].cpu().numpy(), exp_V[a], err_msg="exp_avg_sq" + a)
    assert len(opt.state) == len(RO.PARAMS)                     # the old parameters' entries are gone
    assert model.xys_grad_norm is None and model.vis_counts is None and model.max_2Dsize is None


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
@pytest.mark.parametrize("fused", [False, True], ids=["torch_adam", "fused_adam"])
def test_refinement_equals_reference_golden(path, fused, capsys):
    from bilateral_driving_amd.densify import refinement_after
    from bilateral_driving_amd.optim import FusedAdam
    z = np.load(path)
    ctrl = {k[5:]: z[k].item() for k in z.files if k.startswith("ctrl_")}
    P = {a: z["in" + a] for a in RO.PARAMS}; M = {a: z["in_m" + a] for a in RO.PARAMS}; V = {a: z["in_v" + a] for a in RO.PARAMS}
    stats = {k: z["in_" + k] for k in ("xys_grad_norm", "vis_counts", "max_2Dsize")}
    step = int(z["step"])
    model, opt = build_model(P, M, V, stats, ctrl, float(z["scene_scale"]), int(z["num_train_images"]), step,
                             FusedAdam if fused else torch.optim.Adam)
    refinement_after(model, step, opt, samples=torch.from_numpy(z["samples"]))
    check(model, opt, {a: z["out" + a] for a in RO.PARAMS}, {a: z["out_m" + a] for a in RO.PARAMS},
          {a: z["out_v" + a] for a in RO.PARAMS})
    out = capsys.readouterr().out
    assert f"left points: {z['out_means'].shape[0]}" in out
    # the optimiser keeps working on the new set
    for a in RO.PARAMS:
        getattr(model, a).grad = torch.full_like(getattr(model, a), 1e-3)
    opt.step()


def synthetic(N, seed):
    g = np.random.default_rng(seed)
    r = lambda *s: g.random(s, dtype=np.float32)
    ls = r(N, 3) * 7.5 - 4.5
    small = r(N) < 0.3
    ls[small] = r(int(small.sum()), 3) * 1.6 - 4.5
    lg = r(N, 1) * 9 - 6.5
    # keep every decision away from its threshold (libm and the device differ in the last bit of exp / log)
    for t in (math.log(0.06), math.log(15.0)):
        for tt in (t, t + math.log(1.6)):
            ls = np.where(np.abs(ls - tt) < 2e-3, ls + 5e-3, ls).astype(np.float32)
    lg = np.where(np.abs(lg - math.log(0.005 / 0.995)) < 2e-3, lg + 5e-3, lg).astype(np.float32)
    P = {"_means": (r(N, 3) - 0.5) * 40, "_features_dc": r(N, 3), "_features_rest": (r(N, 15, 3) - 0.5) * 0.2, "_opacities": lg,
         "_scales": ls, "_quats": g.standard_normal((N, 4)).astype(np.float32)}
    M = {a: (g.standard_normal(v.shape) * 1e-2).astype(np.float32) for a, v in P.items()}
    V = {a: (g.random(v.shape) * 1e-4).astype(np.float32) for a, v in P.items()}
    stats = {"xys_grad_norm": r(N) * 0.004, "vis_counts": np.floor(r(N) * 6).astype(np.float32) + 1, "max_2Dsize": r(N) * 0.2}
    return P, M, V, stats


CTRL = dict(warmup_steps=500, reset_alpha_interval=3000, refine_interval=100, n_split_samples=2, reset_alpha_value=0.01,
            densify_grad_thresh=0.0003, densify_size_thresh=0.002, cull_alpha_thresh=0.005, cull_scale_thresh=0.5, cull_screen_size=0.15,
            split_screen_size=0.05, stop_screen_size_at=4000, stop_split_at=15000, sh_degree=3)


@pytest.mark.parametrize("N,step", [(600_001, 3300), (300_000, 1300), (262_145, 16300), (1000, 3100), (255, 3300), (1, 3300)])
def test_refinement_equals_oracle_at_scale(N, step):
    """Sizes on both sides of the 1024-segment boundary of the count scan (N / 256 workgroups); SH degree 3 rows (45 floats)."""
    from bilateral_driving_amd.densify import refinement_after, plan
    P, M, V, stats = synthetic(N, seed=N + step)
    model, opt = build_model(P, M, V, stats, CTRL, 30.0, 150, step, torch.optim.Adam)
    sch = RO.schedule(step, CTRL, 30.0, 150)
    n_split = 0
    if sch["do_densify"]:
        n_split = int(RO.plan(sch, CTRL, P["_scales"], P["_opacities"], stats["xys_grad_norm"], stats["vis_counts"], stats["max_2Dsize"])[0].sum())
    samples = np.random.default_rng(1).standard_normal((2 * n_split, 3)).astype(np.float32)
    eP, eM, eV, ns = RO.refine(step, CTRL, 30.0, 150, P, M, V, stats["xys_grad_norm"], stats["vis_counts"], stats["max_2Dsize"], samples)
    assert ns == n_split
    refinement_after(model, step, opt, samples=torch.from_numpy(samples), verbose=False)
    # a split child is mean + R (scale * noise) with scale up to e^3 and |noise| up to ~5: the sum cancels, so the last-bit
    # differences of exp() between libm and the device are relative to the offset (up to ~100), not to the result
    check(model, opt, eP, eM, eV, means_atol=5e-5)
    if N > 1000 and sch["do_densify"]:
        assert n_split > N // 10 and eP["_means"].shape[0] != N


def test_plan_counts_and_ranks_are_exclusive_scans():
    from bilateral_driving_amd.densify import plan
    N = 300_123
    P, M, V, stats = synthetic(N, seed=5)
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)
    flags, ranks, totals = plan(t(P["_scales"]), t(P["_opacities"]), t(stats["xys_grad_norm"]), t(stats["vis_counts"]), t(stats["max_2Dsize"]),
                                do_densify=True, grad_thresh=0.0003, size_thresh=0.06, split_by_screen=True, split_screen_size=0.05,
                                do_cull=True, cull_alpha_thresh=0.005, cull_by_scale=True, cull_scale_thresh=15.0, cull_by_screen=True,
                                cull_screen_size=0.15)
    f = flags.cpu().numpy(); r = ranks.cpu().numpy().astype(np.int64); tot = totals.cpu().numpy()
    for k, bit in enumerate((0, 2, 3, 4)):
        m = (f >> bit) & 1
        np.testing.assert_array_equal(r[:, k], np.cumsum(m) - m)
    np.testing.assert_array_equal(tot, [int(((f >> b) & 1).sum()) for b in range(5)])
    assert not np.any((f & 8) & ~((f & 1) << 3)) and not np.any((f & 16) & ~((f & 2) << 3))   # kept children imply the parent flag


# ---- node classes: RigidNodes / DeformableNodes.refinement_after (models/nodes/rigid.py:194-325) -------------------------------
RIGID = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refine_rigid_*.npz")))


def attach_instances(model, point_ids, instances_size, dev="cuda"):
    model.point_ids = torch.from_numpy(np.ascontiguousarray(point_ids)).to(dev)
    model.instances_size = torch.from_numpy(np.ascontiguousarray(instances_size)).to(dev)


@pytest.mark.parametrize("path", RIGID, ids=[os.path.basename(f)[:-4] for f in RIGID])
def test_node_refinement_equals_reference_golden(path, capsys):
    """point_ids travel with their rows; with cull_out_of_bound the children of a split are judged by their own sampled mean."""
    from bilateral_driving_amd.densify import refinement_after
    z = np.load(path)
    ctrl = {k[5:]: z[k].item() for k in z.files if k.startswith("ctrl_")}
    P = {a: z["in" + a] for a in RO.PARAMS}; M = {a: z["in_m" + a] for a in RO.PARAMS}; V = {a: z["in_v" + a] for a in RO.PARAMS}
    stats = {k: z["in_" + k] for k in ("xys_grad_norm", "vis_counts", "max_2Dsize")}
    step = int(z["step"])
    model, opt = build_model(P, M, V, stats, ctrl, float(z["scene_scale"]), int(z["num_train_images"]), step, torch.optim.Adam)
    attach_instances(model, z["in_point_ids"], z["instances_size"])
    refinement_after(model, step, opt, samples=torch.from_numpy(z["samples"]))
    check(model, opt, {a: z["out" + a] for a in RO.PARAMS}, {a: z["out_m" + a] for a in RO.PARAMS},
          {a: z["out_v" + a] for a in RO.PARAMS})
    assert model.point_ids.dtype == torch.int64 and model.point_ids.is_cuda
    np.testing.assert_array_equal(model.point_ids.cpu().numpy(), z["out_point_ids"])
    out = capsys.readouterr().out
    assert f"left points: {z['out_means'].shape[0]}" in out
    for a in RO.PARAMS:
        getattr(model, a).grad = torch.full_like(getattr(model, a), 1e-3)
    opt.step()


@pytest.mark.parametrize("N,step,oob", [(200_003, 3300, True), (100_000, 16300, True), (50_000, 1300, False), (3, 3300, True)])
def test_node_refinement_equals_oracle_at_scale(N, step, oob):
    from bilateral_driving_amd.densify import refinement_after
    P, M, V, stats = synthetic(N, seed=N + step + 1)
    g = np.random.default_rng(N)
    n_inst = 37
    ids = g.integers(0, n_inst, size=(N, 1)).astype(np.int64)
    sizes = (g.random((n_inst, 3), dtype=np.float32) * 30 + 8).astype(np.float32)
    # keep the box decision of the ORIGINALS away from its threshold (children: the tolerance below)
    half = sizes[ids[:, 0]] / 2
    near = np.abs(np.abs(P["_means"]) - half) < 1e-3
    P["_means"] = np.where(near, P["_means"] * 0.9, P["_means"]).astype(np.float32)
    ctrl = dict(CTRL, cull_out_of_bound=oob)
    model, opt = build_model(P, M, V, stats, ctrl, 30.0, 150, step, torch.optim.Adam)
    attach_instances(model, ids, sizes)
    sch = RO.schedule(step, ctrl, 30.0, 150)
    n_split = 0
    if sch["do_densify"]:
        n_split = int(RO.plan(sch, ctrl, P["_scales"], P["_opacities"], stats["xys_grad_norm"], stats["vis_counts"], stats["max_2Dsize"])[0].sum())
    samples = np.random.default_rng(2).standard_normal((2 * n_split, 3)).astype(np.float32)
    if oob and n_split:
        # a sampled child within float rounding of a box face could fall on the other side on the device (last bits of exp and of
        # the rotation): pull the noise of such children in until no new row sits within 1e-3 of a face
        split, dup, keep_o, keep_s, keep_d = RO.plan(sch, ctrl, P["_scales"], P["_opacities"], stats["xys_grad_norm"],
                                                     stats["vis_counts"], stats["max_2Dsize"])
        r_split = np.cumsum(split) - split
        parents = np.nonzero(keep_s)[0]
        KO, KS = int(keep_o.sum()), int(keep_s.sum())
        for _ in range(8):
            nP, _, _, _, nids = RO.refine_nodes(step, dict(ctrl, cull_out_of_bound=False), 30.0, 150, P, M, V, stats["xys_grad_norm"],
                                                stats["vis_counts"], stats["max_2Dsize"], samples, ids, sizes)
            near = (np.abs(np.abs(nP["_means"]) - sizes[nids[:, 0]] / 2) < 1e-3).any(axis=-1)
            rows = np.nonzero(near[KO:KO + 2 * KS])[0]
            if rows.size == 0:
                break
            samples[(rows // KS) * n_split + r_split[parents[rows % KS]]] *= np.float32(0.7)
        else:
            raise AssertionError("could not move the children away from the box faces")
    eP, eM, eV, ns, eids = RO.refine_nodes(step, ctrl, 30.0, 150, P, M, V, stats["xys_grad_norm"], stats["vis_counts"],
                                           stats["max_2Dsize"], samples, ids, sizes)
    refinement_after(model, step, opt, samples=torch.from_numpy(samples), verbose=False)
    got_n = model._means.shape[0]
    check(model, opt, eP, eM, eV, means_atol=5e-5)
    np.testing.assert_array_equal(model.point_ids.cpu().numpy(), eids)
    if N > 1000 and oob:
        assert got_n < N + 2 * n_split


def test_out_of_bound_mask_matches_reference_expression():
    from bilateral_driving_amd.densify import out_of_bound_mask
    g = torch.Generator().manual_seed(3)
    N, I = 70_001, 11
    means = ((torch.rand(N, 3, generator=g) - 0.5) * 40).cuda()
    ids = torch.randint(0, I, (N, 1), generator=g).cuda()
    sizes = (torch.rand(I, 3, generator=g) * 30 + 8).cuda()
    mask = out_of_bound_mask(means, ids, sizes)
    ref = (means.abs() > sizes[ids[..., 0]] / 2).any(dim=-1)            # nodes/rigid.py:379-382
    assert mask.dtype == torch.uint8 and torch.equal(mask.bool(), ref)
    assert 0 < int(ref.sum()) < N
    bad = ids.clone(); bad[5] = I; bad[9] = -1                           # ids outside the table: reported out of bound, no fault
    m2 = out_of_bound_mask(means, bad, sizes)
    assert bool(m2[5]) and bool(m2[9])


def test_plan_extra_cull_only_touches_originals():
    from bilateral_driving_amd.densify import plan
    N = 10_000
    P, M, V, stats = synthetic(N, seed=9)
    t = lambda a: torch.from_numpy(a).cuda()
    kw = dict(do_densify=True, grad_thresh=0.0003, size_thresh=0.06, split_by_screen=True, split_screen_size=0.05, do_cull=True,
              cull_alpha_thresh=0.005, cull_by_scale=True, cull_scale_thresh=15.0, cull_by_screen=True, cull_screen_size=0.15)
    args = (t(P["_scales"]), t(P["_opacities"]), t(stats["xys_grad_norm"]), t(stats["vis_counts"]), t(stats["max_2Dsize"]))
    f0, _, t0 = plan(*args, **kw)
    extra = (torch.arange(N, device="cuda") % 3 == 0).to(torch.uint8)
    f1, _, t1 = plan(*args, **kw, extra_cull=extra)
    f0, f1 = f0.cpu().numpy(), f1.cpu().numpy()
    e = extra.cpu().numpy().astype(bool)
    np.testing.assert_array_equal(f1 & ~np.uint8(4), f0 & ~np.uint8(4))                  # split / dup / children bits unchanged
    np.testing.assert_array_equal((f1 >> 2) & 1, ((f0 >> 2) & 1) & ~e)
    assert int(t1[2]) < int(t0[2]) and int(t1[3]) == int(t0[3]) and int(t1[4]) == int(t0[4])


@pytest.mark.parametrize("nodes", [False, True])
def test_refinement_that_culls_everything_leaves_empty_tensors(nodes):
    from bilateral_driving_amd.densify import refinement_after
    N, step = 300, 16300
    P, M, V, stats = synthetic(N, seed=2)
    P["_opacities"][:] = -20.0                                            # sigmoid << cull_alpha_thresh
    ctrl = dict(CTRL, cull_out_of_bound=nodes)
    model, opt = build_model(P, M, V, stats, ctrl, 30.0, 150, step, torch.optim.Adam)
    if nodes:
        attach_instances(model, np.zeros((N, 1), np.int64), np.full((1, 3), 100.0, np.float32))
    refinement_after(model, step, opt, verbose=False)
    for a in RO.PARAMS:
        prm = getattr(model, a)
        assert prm.shape[0] == 0 and prm.shape[1:] == torch.Size(P[a].shape[1:])
        assert opt.state[prm]["exp_avg"].shape == prm.shape
    if nodes:
        assert model.point_ids.shape == (0, 1) and model.point_ids.dtype == torch.int64


def test_refinement_with_reorder_leaves_the_same_set_in_spatial_order():
    """``refinement_after(..., reorder=True)``: the same Gaussians, Adam moments and group wiring as without, rows in
    ``spatial_order`` of the new centres (the order carries no meaning in the reference: vanilla.py:256-262 appends children)."""
    from bilateral_driving_amd.densify import refinement_after, spatial_order
    from bilateral_driving_amd.optim import FusedAdam
    N, step = 40_000, 3300
    P, M, V, stats = synthetic(N, seed=5)
    outs = []
    for reorder in (False, True):
        model, opt = build_model(P, M, V, stats, CTRL, 30.0, 150, step, FusedAdam)
        g = torch.Generator().manual_seed(9)
        refinement_after(model, step, opt, verbose=False, reorder=reorder,
                         sample_fn=lambda shape, dev: torch.randn(shape, generator=g).to(dev))
        outs.append((model, opt))
    (ma, oa), (mb, ob) = outs
    n = ma._means.shape[0]
    assert mb._means.shape[0] == n and n != N
    perm = spatial_order(ma._means)
    assert torch.equal(spatial_order(mb._means), torch.arange(n, device="cuda"))            # already in order
    for a, name in zip(RO.PARAMS, GROUPS):
        assert torch.equal(getattr(mb, a).detach(), getattr(ma, a).detach()[perm]), a
        assert torch.equal(ob.state[getattr(mb, a)]["exp_avg_sq"], oa.state[getattr(ma, a)]["exp_avg_sq"][perm]), a
        grp = [gr for gr in ob.param_groups if gr["name"] == mb.class_prefix + name][0]
        assert grp["params"][0] is getattr(mb, a)


def test_deferred_row_adam_goes_through_a_refinement_bit_equal_to_the_dense_one():
    """optim.DeferredRowAdam keeps the SH rows of unseen Gaussians behind; ``before_refinement()`` brings them up to date, the
    refinement's optimizer-state surgery (basics.py:162-206 as densify._move_rows does it) moves parameters and moments, the rows'
    step words are made again at the new size -- and the run continues.  Same random visibility, same gradients, same refinement:
    the SH parameters and both moments equal the dense FusedAdam's bit for bit before and after the refinement."""
    from bilateral_driving_amd.densify import refinement_after
    from bilateral_driving_amd.optim import DeferredRowAdam, FusedAdam
    N, step = 20_011, 3300
    P, M, V, stats = synthetic(N, seed=5)
    sch = RO.schedule(step, CTRL, 30.0, 150)
    n_split = int(RO.plan(sch, CTRL, P["_scales"], P["_opacities"], stats["xys_grad_norm"], stats["vis_counts"], stats["max_2Dsize"])[0].sum())
    samples = torch.from_numpy(np.random.default_rng(1).standard_normal((2 * n_split, 3)).astype(np.float32))
    deferred = ("_features_dc", "_features_rest")

    def make(lazy):
        model, opt0 = build_model(P, M, V, stats, CTRL, 30.0, 150, step, FusedAdam)
        groups = []
        for a, n in zip(RO.PARAMS, GROUPS):
            g = {"params": [getattr(model, a)], "lr": 2e-3, "eps": 1e-15, "weight_decay": 0, "name": model.class_prefix + n}
            if lazy and a in deferred:
                g["deferred_rows"] = True
            groups.append(g)
        opt = (DeferredRowAdam if lazy else FusedAdam)(groups, lr=0.0, eps=1e-15)
        for a in RO.PARAMS:      # the moments of build_model (a run in progress)
            prm = getattr(model, a)
            st = opt.state[prm]
            st["exp_avg"], st["exp_avg_sq"] = opt0.state[prm]["exp_avg"].clone(), opt0.state[prm]["exp_avg_sq"].clone()
            if "step" not in st:
                st["step"] = torch.tensor(0.0)
        return model, opt

    def steps(model, opt, lazy, k0, k1):
        for it in range(k0, k1):
            gen = torch.Generator().manual_seed(1000 + it)
            n = model._means.shape[0]
            ids = torch.randperm(n, generator=gen)[: max(n // 7, 1)].sort().values
            for a in RO.PARAMS:
                prm = getattr(model, a)
                g = torch.zeros(prm.shape)
                g[ids] = torch.randn((ids.numel(),) + tuple(prm.shape[1:]), generator=gen) * 0.01     # dense zeros elsewhere
                prm.grad = g.cuda()
            if lazy:
                lst = ids.to(torch.int32).cuda()
                opt.catchup(lst.numel(), None, lst)               # the forward: the listed rows are made current before they are read
                opt.step(lists=[(lst.numel(), None, lst)])
            else:
                opt.step()

    (ma, oa), (mb, ob) = make(False), make(True)
    steps(ma, oa, False, 0, 12); steps(mb, ob, True, 0, 12)
    assert int((ob.state[mb._features_rest]["last_step"] < 12).sum()) > 0          # rows no list has named are behind
    ob.before_refinement()
    for a in deferred:
        assert torch.equal(getattr(ma, a).detach(), getattr(mb, a).detach()), a
    for m_, o_ in ((ma, oa), (mb, ob)):
        for k, v in stats.items():
            setattr(m_, k, torch.from_numpy(v).cuda())
        refinement_after(m_, step, o_, samples=samples, verbose=False)
    assert ma._means.shape[0] == mb._means.shape[0] != N
    steps(ma, oa, False, 12, 24); steps(mb, ob, True, 12, 24)
    ob.flush()
    for a in RO.PARAMS:
        pa, pb = getattr(ma, a), getattr(mb, a)
        assert torch.equal(pa.detach(), pb.detach()), a
        assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]) and torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), a
    assert int(ob.state[mb._features_rest]["last_step"].min()) == 24 and ob.state[mb._features_rest]["last_step"].shape[0] == mb._means.shape[0]
