"""-m gpu: the image losses at RANDOM sizes against oracle/loss_oracle.py in float64 -- SSIM (pytorch_msssim 1.0.0 restated: 11-tap
Gaussian window, valid region; trainers/base.py:557-563) from the smallest image the window fits, strips one window wide, sizes that
are not multiples of anything; the regularisers' fused node (trainers/base.py:566-585,638-659) with every subset of its terms.
The fixed sizes live in tests/test_gpu_00 / test_gpu_06; the pixel loss is pinned by goldens of the reference's own models/losses.py.

BDS_SWEEP_CASES (default 10) cases each."""
import os

import pytest
import torch

from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("BDS_SWEEP_CASES", "10"))


def _size(g, lo):
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
    kind = ri(0, 3)
    if kind == 0:
        return ri(lo, lo + 6), ri(lo, lo + 6)
    if kind == 1:
        return (lo, ri(lo, 500)) if ri(0, 1) else (ri(lo, 400), lo)
    return ri(lo, 300), ri(lo, 420)


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_ssim_random_size(seed):
    from bilateral_driving_amd.losses import ssim, ssim_loss
    g = torch.Generator().manual_seed(11000 + seed)
    H, W = _size(g, 11)
    gt = torch.rand(H, W, 3, generator=g)
    noise = (0.02, 0.15, 0.6)[seed % 3]
    pred = (gt + noise * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    p_ref = pred.clone().double().requires_grad_(True)
    s_ref = LO.ssim(gt.double(), p_ref)
    (1.7 * (1 - s_ref)).backward()
    p = pred.cuda().requires_grad_(True)
    s = ssim(p, gt.cuda())
    (1.7 * ssim_loss(p, gt.cuda())).backward()
    assert abs(float(s) - float(s_ref)) < 2e-5, (H, W, float(s), float(s_ref))
    gref = p_ref.grad.float()
    assert float((p.grad.cpu() - gref).abs().max()) <= 1e-4 * float(gref.abs().max()) + 1e-9, (H, W)


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_reg_losses_random_size_and_terms(seed):
    from bilateral_driving_amd import losses as Ls
    g = torch.Generator().manual_seed(12000 + seed)
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
    H, W = _size(g, 1)
    pix = torch.rand(H, W, 3, generator=g)
    rgb = torch.rand(H, W, 3, generator=g) * 1.1
    op = torch.rand(H, W, 1, generator=g)
    op[0, 0, 0], op[-1, -1, 0] = 0.0, 1.0
    dep = torch.rand(H, W, 1, generator=g) * 30 + 0.2
    dyn = torch.rand(H, W, 1, generator=g) * (0.1 if ri(0, 4) == 0 else 1.0)
    egocar = (torch.rand(H, W, generator=g) > 0.8).float() if ri(0, 1) else None
    use_op, use_dyn = bool(ri(0, 1)), bool(ri(0, 1))
    use_dep = bool(ri(0, 1)) and H > 1 and W > 1
    if not (use_op or use_dyn or use_dep):
        use_op = True
    w = torch.rand(3, generator=g) + 0.2
    o64 = {k: v.double().requires_grad_(True) for k, v in dict(op=op, dep=dep, rgb=rgb).items()}
    t_ref = LO.reg_losses(pix.double(), o64["op"] if use_op else None, o64["dep"] if use_dep else None, o64["rgb"],
                          dyn.double() if use_dyn else None, None if egocar is None else egocar.double())
    if t_ref.requires_grad:      # (only the dynamic-region term, and nothing above its threshold: a constant zero)
        (t_ref * w.double()).sum().backward()
    c = {k: v.cuda().requires_grad_(True) for k, v in dict(op=op, dep=dep, rgb=rgb).items()}
    t = Ls.reg_losses(pix.cuda(), c["op"] if use_op else None, c["dep"] if use_dep else None, c["rgb"], dyn.cuda() if use_dyn else None,
                      None if egocar is None else egocar.cuda())
    if t.requires_grad:
        (t * w.cuda()).sum().backward()
    tag = (H, W, use_op, use_dep, use_dyn, egocar is not None)
    assert torch.allclose(t.cpu().double(), t_ref.detach(), rtol=2e-5, atol=1e-7), (tag, t, t_ref)
    for k in ("op", "dep", "rgb"):
        ref, got = o64[k].grad, c[k].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert got is None or float(got.abs().max()) == 0.0, (k, tag)
            continue
        assert float((got.cpu().double() - ref).norm() / ref.norm()) < 2e-5, (k, tag)
