"""-m gpu: the regularisers of the reference's image loss (models/trainers/base.py:566-585, 638-659) as one fused node against
oracle/loss_oracle.py::reg_losses (float64 transcription of the trainer's expressions; the inverse-depth smoothness term is
kornia's, an external package absent here: parity unpinned for that term)."""
import pytest
import torch

from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,ego,terms", [(37, 53, True, "all"), (270, 480, False, "all"), (16, 1, False, "entropy"), (9, 31, True, "dyn"),
                                          (64, 48, False, "nodyn_pixels")])
def test_reg_losses_match_oracle(H, W, ego, terms):
    from bilateral_driving_amd import losses as Ls
    g = torch.Generator().manual_seed(H * 131 + W)
    pix = torch.rand(H, W, 3, generator=g)
    rgb = torch.rand(H, W, 3, generator=g) * 1.1
    op = torch.rand(H, W, 1, generator=g)
    op[0, 0, 0], op[-1, -1, 0] = 0.0, 1.0                 # outside the clamp interval: zero gradient
    dep = torch.rand(H, W, 1, generator=g) * 30 + 0.2
    dyn = torch.rand(H, W, 1, generator=g)
    if terms == "nodyn_pixels":
        dyn = dyn * 0.1                                   # nothing above the 0.2 threshold: the term vanishes
    egocar = (torch.rand(H, W, generator=g) > 0.8).float() if ego else None
    use_op = terms in ("all", "entropy", "nodyn_pixels")
    use_dep = terms in ("all", "nodyn_pixels") and H > 1 and W > 1
    use_dyn = terms in ("all", "dyn", "nodyn_pixels")
    w = torch.tensor([0.7, 1.3, 0.4])
    # oracle (float64)
    o64 = {k: v.double().requires_grad_(True) for k, v in dict(op=op, dep=dep, rgb=rgb).items()}
    t_ref = LO.reg_losses(pix.double(), o64["op"] if use_op else None, o64["dep"] if use_dep else None, o64["rgb"],
                          dyn.double() if use_dyn else None, None if egocar is None else egocar.double())
    (t_ref * w.double()).sum().backward()
    # kernels
    c = {k: v.cuda().requires_grad_(True) for k, v in dict(op=op, dep=dep, rgb=rgb).items()}
    t = Ls.reg_losses(pix.cuda(), c["op"] if use_op else None, c["dep"] if use_dep else None, c["rgb"], dyn.cuda() if use_dyn else None,
                      None if egocar is None else egocar.cuda())
    (t * w.cuda()).sum().backward()
    assert torch.allclose(t.cpu().double(), t_ref.detach(), rtol=2e-5, atol=1e-7), (t, t_ref)
    if terms == "nodyn_pixels":
        assert float(t[2]) == 0.0
    for k in ("op", "dep", "rgb"):
        ref, got = o64[k].grad, c[k].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert got is None or float(got.abs().max()) == 0.0, k
            continue
        assert float((got.cpu().double() - ref).norm() / ref.norm()) < 2e-5, k
    if use_op:
        assert float(c["op"].grad[0, 0, 0]) == 0.0 and float(c["op"].grad[-1, -1, 0]) == 0.0
