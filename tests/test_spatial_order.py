"""Host logic of the spatial row order (``densify.spatial_order`` / ``reorder_rows``): a permutation, neighbours in memory are
neighbours in space, and everything that belongs to a row -- parameters, Adam moments in their named groups
(/root/reference/project/models/trainers/base.py:201-222), ``point_ids``, the densification statistics -- moves with it."""
import types

import torch

from bilateral_driving_amd.densify import _ATTRS, _GROUPS, reorder_rows, spatial_order


def test_spatial_order_is_a_permutation_that_clusters_rows():
    g = torch.Generator().manual_seed(0)
    m = torch.rand(5000, 3, generator=g) * torch.tensor([80.0, 80.0, 8.0])
    p = spatial_order(m)
    assert p.dtype == torch.int64 and sorted(p.tolist()) == list(range(5000))
    step_before = (m[1:] - m[:-1]).norm(dim=1).mean()
    step_after = (m[p][1:] - m[p][:-1]).norm(dim=1).mean()
    assert step_after < 0.25 * step_before
    assert torch.equal(spatial_order(m[p]), torch.arange(5000))          # idempotent (stable sort of equal codes)
    assert spatial_order(torch.zeros(1, 3)).tolist() == [0] and spatial_order(torch.ones(7, 3)).tolist() == list(range(7))


def test_reorder_rows_moves_parameters_state_ids_and_statistics_together():
    N = 300
    g = torch.Generator().manual_seed(1)
    shapes = {"_means": (N, 3), "_features_dc": (N, 3), "_features_rest": (N, 8, 3), "_opacities": (N, 1), "_scales": (N, 3), "_quats": (N, 4)}
    model = types.SimpleNamespace(class_prefix="Background#")
    groups = []
    for a, n in zip(_ATTRS, _GROUPS):
        prm = torch.nn.Parameter(torch.randn(shapes[a], generator=g))
        setattr(model, a, prm)
        groups.append({"params": [prm], "lr": 1e-3, "name": model.class_prefix + n})
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for a in _ATTRS:
        getattr(model, a).grad = torch.randn(shapes[a], generator=g)
    opt.step()
    model.point_ids = torch.arange(N)[:, None].clone()
    model.xys_grad_norm, model.vis_counts, model.max_2Dsize = torch.rand(N, generator=g), torch.ones(N), torch.rand(N, generator=g)
    tag = model._means.detach()[:, 0].clone()                              # identifies a row through the move
    old = {a: getattr(model, a).detach().clone() for a in _ATTRS}
    old_m = {a: opt.state[getattr(model, a)]["exp_avg"].clone() for a in _ATTRS}
    old_x = model.xys_grad_norm.clone()
    perm = spatial_order(model._means)
    reorder_rows(model, opt, perm)
    assert torch.equal(model._means.detach()[:, 0], tag[perm])
    for a, n in zip(_ATTRS, _GROUPS):
        prm = getattr(model, a)
        assert isinstance(prm, torch.nn.Parameter) and torch.equal(prm.detach(), old[a][perm])
        grp = [gr for gr in opt.param_groups if gr["name"] == model.class_prefix + n][0]
        assert grp["params"] == [prm] or (len(grp["params"]) == 1 and grp["params"][0] is prm)
        assert torch.equal(opt.state[prm]["exp_avg"], old_m[a][perm]) and float(opt.state[prm]["step"]) == 1.0
    assert len(opt.state) == len(_ATTRS)
    assert torch.equal(model.point_ids[:, 0], perm) and torch.equal(model.xys_grad_norm, old_x[perm])
    # the optimizer keeps working on the moved rows
    for a in _ATTRS:
        getattr(model, a).grad = torch.zeros(shapes[a])
    opt.step()
