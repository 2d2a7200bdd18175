"""oracle/color_correct_oracle.py (masked normal equations) against the reference's own color_correct (tall least squares):
tests/golden/color_correct_*.npz, produced by oracle/gen_golden_color_correct.py."""
import glob
import os

import numpy as np
import pytest

from oracle.color_correct_oracle import color_correct

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "color_correct_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_equals_reference(path):
    z = np.load(path)
    out = color_correct(z["img"], z["ref"], int(z["num_iters"]))
    assert out.shape == z["img"].shape
    np.testing.assert_allclose(out, z["out_f64"], atol=2e-9)
    np.testing.assert_allclose(out, z["out_f32"], atol=5e-6)
    assert np.abs(out - z["ref"]).mean() < 0.5 * np.abs(z["img"] - z["ref"]).mean()      # it does correct the colours


def test_two_goldens_present():
    assert len(FILES) == 2
