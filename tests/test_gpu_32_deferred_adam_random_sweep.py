"""-m gpu: optim.DeferredRowAdam against the dense FusedAdam pass, BIT FOR BIT, on RANDOM schedules: 1 to 20 k rows of 1 / 4 / 9 / 16
SH bands, lists from empty to every row (above ``dense_above`` the step takes the dense pass), two lists naming the same rows, steps
without a view, ring tables of 8 ... 1024 steps (the wrap flushes), weight decay, learning rates rewritten every step, a mid-run
flush (what refinement and checkpoints do).  The fixed cases: tests/test_gpu_10.  The reference's cadence: one dense Adam step per
single-view iteration (models/trainers/base.py:222-226,502-516).

BDS_SWEEP_CASES (default 8) cases."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("BDS_SWEEP_CASES", "8"))


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_deferred_row_adam_random_schedule(seed):
    from bilateral_driving_amd.optim import DeferredRowAdam, FusedAdam
    gen = torch.Generator().manual_seed(32000 + seed)
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=gen))
    N = int(10 ** (float(torch.rand(1, generator=gen)) * 4.3))
    K = (1, 4, 9, 16)[ri(0, 3)]
    steps, table_steps, wd = ri(8, 70), (8, 16, 64, 1024)[ri(0, 3)], (0.0, 0.01)[ri(0, 1)]
    dense_above = (0.0, 0.3, 0.6, 1.1)[ri(0, 3)]
    sh0 = torch.randn(N, K, 3, generator=gen).cuda()
    dc = sh0[:, :1].clone().contiguous().requires_grad_(True)
    rest = sh0[:, 1:].clone().contiguous().requires_grad_(True) if K > 1 else None
    ga = [{"params": [dc], "lr": 2.5e-3, "weight_decay": wd}] + ([{"params": [rest], "lr": 2.5e-3 / 20, "weight_decay": wd}] if K > 1 else [])
    oa = FusedAdam(ga, lr=0.0, eps=1e-15)
    sh = sh0.clone().requires_grad_(True)
    ob = DeferredRowAdam([{"params": [sh], "lr": 2.5e-3, "lr_b": 2.5e-3 / 20, "col_split": 3, "deferred_rows": True, "weight_decay": wd}],
                         lr=0.0, eps=1e-15, table_steps=table_steps, dense_above=dense_above)
    sh.grad = torch.zeros_like(sh)

    def make_list():
        kind = ri(0, 5)
        n = 0 if kind == 0 else (N if kind == 1 else ri(0, N))
        ids = torch.randperm(N, generator=gen)[:n].sort().values.to(torch.int32)
        cap = n + ri(1, 7)                       # the list's capacity (what ``dense_above`` adds up): a few slots of padding
        pad = torch.full((cap,), -1, dtype=torch.int32)
        pad[:n] = ids
        return pad.cuda(), torch.tensor([n], dtype=torch.int64).cuda(), ids.long().cuda(), cap

    def same():
        a = torch.equal(sh.detach()[:, :1], dc.detach())
        return a and (K == 1 or torch.equal(sh.detach()[:, 1:], rest.detach()))

    flush_at = ri(1, steps - 1)
    for it in range(steps):
        lists = [make_list() for _ in range(ri(0, 2))]
        for ids, cnt, rows, cap in lists:
            ob.catchup(cap, cnt.data_ptr(), ids)
            assert torch.equal(sh.detach()[rows, :1], dc.detach()[rows]) and (K == 1 or torch.equal(sh.detach()[rows, 1:], rest.detach()[rows])), (seed, it)
        g = torch.zeros(N, K, 3, device="cuda")
        for ids, cnt, rows, cap in lists:
            g[rows] += torch.randn(rows.numel(), K, 3, generator=gen).cuda() * 0.01
        dc.grad = g[:, :1].contiguous()
        if K > 1:
            rest.grad = g[:, 1:].contiguous()
        sh.grad.copy_(g)
        for grp in oa.param_groups:
            grp["lr"] *= 0.98
        ob.param_groups[0]["lr"] *= 0.98; ob.param_groups[0]["lr_b"] *= 0.98
        oa.step()
        ob.step(lists=[(cap, cnt.data_ptr(), ids) for ids, cnt, rows, cap in lists])
        if it == flush_at:
            ob.flush()
            assert same(), (seed, it, "flush")
    ob.flush()
    tag = dict(seed=seed, N=N, K=K, steps=steps, table_steps=table_steps, wd=wd, dense_above=dense_above)
    assert same(), tag
    assert torch.equal(ob.state[sh]["exp_avg"][:, :1], oa.state[dc]["exp_avg"]) and torch.equal(ob.state[sh]["exp_avg_sq"][:, :1], oa.state[dc]["exp_avg_sq"]), tag
    if K > 1:
        assert torch.equal(ob.state[sh]["exp_avg"][:, 1:], oa.state[rest]["exp_avg"]) and torch.equal(ob.state[sh]["exp_avg_sq"][:, 1:], oa.state[rest]["exp_avg_sq"]), tag
