"""Colour-correct post-process on the device (bilateral_driving_amd.colorcorrect -> csrc/colorcorrect.hip through the C ABI) against
the reference's own outputs (goldens) and, at 1080p, against the pinned oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.color_correct_oracle import color_correct as oracle_cc

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "color_correct_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_equals_reference_golden(path):
    from bilateral_driving_amd.colorcorrect import color_correct
    z = np.load(path)
    img, ref = torch.from_numpy(z["img"]).float().cuda(), torch.from_numpy(z["ref"]).float().cuda()
    out = color_correct(img, ref, int(z["num_iters"]))
    assert out.shape == img.shape and out.dtype == torch.float32
    got = out.cpu().numpy()
    np.testing.assert_allclose(got, z["out_f64"], atol=2e-5)       # float32 image arithmetic, double accumulation and solve
    np.testing.assert_allclose(got, z["out_f32"], atol=2e-5)
    assert torch.equal(img, torch.from_numpy(z["img"]).float().cuda())        # inputs untouched


def test_full_frame_equals_oracle_and_matches_identity_cases():
    from bilateral_driving_amd.colorcorrect import color_correct
    H, W = 1080, 1920
    g = torch.Generator().manual_seed(3)
    ref = (torch.rand(H, W, 3, generator=g) * 1.1 - 0.05).clamp(0, 1)
    A = torch.eye(3) + 0.1 * torch.randn(3, 3, generator=g)
    img = (ref @ A.T + 0.05 * ref ** 2 - 0.02 + 0.01 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    out = color_correct(img.cuda(), ref.cuda(), 10).cpu().numpy()
    exp = oracle_cc(img.numpy(), ref.numpy(), 10)
    assert np.abs(out - exp).max() < 5e-5
    assert np.abs(out - ref.numpy()).mean() < 0.7 * np.abs(img.numpy() - ref.numpy()).mean()
    # an image that already equals its reference stays put (up to the least-squares fit of the clipped ends)
    same = color_correct(ref.cuda(), ref.cuda(), 3).cpu()
    assert float((same - ref).abs().max()) < 1e-4
    # zero iterations: the input itself
    assert torch.equal(color_correct(img.cuda(), ref.cuda(), 0).cpu(), img)


def test_argument_errors():
    from bilateral_driving_amd import _lib as L
    from bilateral_driving_amd.colorcorrect import color_correct
    with pytest.raises(ValueError):
        color_correct(torch.rand(4, 4, 3).cuda(), torch.rand(4, 4, 4).cuda())
    with pytest.raises(L.BdsError):
        color_correct(torch.rand(4, 4, 3), torch.rand(4, 4, 3))
