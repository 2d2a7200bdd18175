"""-m gpu: FusedAdam against torch.optim.Adam (the optimiser the reference builds, models/trainers/base.py:222-226) on RANDOM group
sets: 1-30 groups (more than one multi-tensor launch holds), element counts from 1 to 3 M that are and are not multiples of four
(the 16-byte body and its tail), per-group lr / betas / eps / weight decay, parameters without a gradient in some steps, all-zero
gradients, learning rates rewritten between steps, with and without the one-launch path for the small groups
(``BDS_ADAM_MULTI``) and with gradients consumed (cleared) by the step.

BDS_SWEEP_CASES (default 8) cases."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("BDS_SWEEP_CASES", "8"))


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_fused_adam_random_groups(seed, monkeypatch):
    assert torch.cuda.is_available()
    from bilateral_driving_amd import optim as O
    g = torch.Generator().manual_seed(31000 + seed)
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
    ru = lambda a, b: float(torch.rand(1, generator=g)) * (b - a) + a
    monkeypatch.setenv("BDS_ADAM_MULTI", str(seed % 2))
    n_groups = ri(1, 30)
    shapes = []
    for _ in range(n_groups):
        kind = ri(0, 4)
        if kind == 0:
            shapes.append((ri(1, 9),))
        elif kind == 1:
            shapes.append((ri(1, 3000), (3, 4, 1, 45)[ri(0, 3)]))
        elif kind == 2:
            shapes.append((ri(1, 6), 12, ri(1, 8), ri(1, 16), ri(1, 16)))      # bilateral grids
        elif kind == 3:
            shapes.append((int(10 ** ru(3.0, 6.5)),))
        else:
            shapes.append((ri(1, 70000), 3))
    base = [torch.randn(s, generator=g) for s in shapes]
    cfgs = [dict(lr=10 ** ru(-5, -1), betas=((0.9, 0.8, 0.0)[ri(0, 2)], (0.999, 0.99)[ri(0, 1)]), eps=(1e-15, 1e-8)[ri(0, 1)],
                 weight_decay=(0.0, 0.0, 0.01)[ri(0, 2)]) for _ in shapes]
    consume = bool(ri(0, 1))

    def make(cls, **kw):
        ps = [b.clone().cuda().requires_grad_(True) for b in base]
        return ps, cls([dict(params=[p], **c) for p, c in zip(ps, cfgs)], lr=0.0, **kw)

    pa, oa = make(torch.optim.Adam)
    pb, ob = make(O.FusedAdam, consume_grads=consume)
    steps = ri(3, 12)
    for it in range(steps):
        for i, (x, y) in enumerate(zip(pa, pb)):
            r = ri(0, 9)
            if r == 0 and it > 0:          # no gradient this step: both optimisers leave the parameter and its state alone
                x.grad = None; y.grad = None
                continue
            gr = torch.zeros(x.shape) if r == 1 else torch.randn(x.shape, generator=g) * 10.0 ** ri(-3, 1)
            x.grad = gr.cuda(); y.grad = gr.cuda()
        if it == steps // 2:
            for ga, gb in zip(oa.param_groups, ob.param_groups):
                ga["lr"] *= 0.3; gb["lr"] *= 0.3
        oa.step(); ob.step()
        if consume:
            assert all(y.grad is None or float(y.grad.abs().max()) == 0.0 for y in pb)
    for i, (x, y) in enumerate(zip(pa, pb)):
        sa, sb = oa.state[x], ob.state[y]
        assert float(sa["step"]) == float(sb["step"]), (i, shapes[i])
        for a, b, what in ((x, y, "param"), (sa["exp_avg"], sb["exp_avg"], "exp_avg"), (sa["exp_avg_sq"], sb["exp_avg_sq"], "exp_avg_sq")):
            err = float((a.detach() - b.detach()).abs().max()) / max(float(a.detach().abs().max()), 1e-30)
            assert err < 5e-6, (i, shapes[i], cfgs[i], what, err)
