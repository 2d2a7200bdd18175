"""The register dataflow of csrc/mlp_head.hip, modelled lane by lane with the documented v_mfma_f32_32x32x2_f32 operand layout
(oracle/mfma_dataflow_model.py), against plain torch autograd of the reference's head: 3 bias-free Linear layers with tanh
(/root/reference/project/models/modules.py:621-627) and the trainer's application with the residual (scene_graph.py:99-102).
Proves the index maps (chained D -> B operands, the placement of the 12 affine entries, the LDS transposes of the weight
gradients) on the CPU; the kernel itself is compared with torch on the GPU (tests/test_gpu_12_neural_modules.py)."""
import numpy as np
import pytest
import torch

from oracle import mfma_dataflow_model as DM


def reference(feats, rgb, W1, W2, W3, v_out, residual):
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    f, c, w1, w2, w3 = t(feats), t(rgb), t(W1), t(W2), t(W3)
    A = (torch.tanh(torch.tanh(f @ w1.T) @ w2.T) @ w3.T).reshape(-1, 3, 4)
    out = (A[..., :3] @ c[..., None])[..., 0] + A[..., 3]
    if residual:
        out = out + c
    out.backward(torch.tensor(v_out, dtype=torch.float64))
    return out.detach().numpy(), dict(v_feats=f.grad.numpy(), v_rgb=c.grad.numpy(), v_w1=w1.grad.numpy(), v_w2=w2.grad.numpy(),
                                      v_w3=w3.grad.numpy())


def test_d_layout_rowmap_is_a_bijection():
    rows = {(int(DM.rowmap(r, h))) for r in range(16) for h in (0, 1)}
    assert rows == set(range(32))
    # four consecutive registers of a half are four consecutive rows (float4 stores / b128 weight reads)
    for g in range(4):
        for h in (0, 1):
            assert [int(DM.rowmap(4 * g + k, h)) for k in range(4)] == [8 * g + 4 * h + k for k in range(4)]


def test_mfma_model_identity_with_asymmetric_b():
    g = np.random.default_rng(0)
    B = g.standard_normal((2, 32)).astype(np.float32)
    A = np.zeros((32, 2), np.float32); A[0, 0] = 1; A[5, 1] = 2
    d = DM.mfma_32x32x2(A[DM.COL, DM.HALF], B[DM.HALF, DM.COL], DM.zeros_tile())
    M = DM.tile_to_matrix(d)
    np.testing.assert_array_equal(M, A @ B)


@pytest.mark.parametrize("F", [24, 16, 8])
@pytest.mark.parametrize("residual", [True, False])
def test_lane_model_equals_autograd(F, residual):
    g = np.random.default_rng(F)
    feats = g.standard_normal((32, F)).astype(np.float32)
    rgb = g.random((32, 3), dtype=np.float32)
    W1 = (g.standard_normal((64, F)) * 0.3).astype(np.float32)
    W2 = (g.standard_normal((64, 64)) * 0.2).astype(np.float32)
    W3 = (g.standard_normal((12, 64)) * 0.2).astype(np.float32)
    v_out = g.standard_normal((32, 3)).astype(np.float32)
    out, _ = DM.forward(feats, rgb, W1, W2, W3, residual)
    grads = DM.backward(feats, rgb, W1, W2, W3, v_out, residual)
    ref_out, ref = reference(feats, rgb, W1, W2, W3, v_out, residual)
    np.testing.assert_allclose(out, ref_out, rtol=2e-5, atol=2e-5)
    for k, v in ref.items():
        scale = np.abs(v).max()
        assert grads[k].shape == v.shape, k
        np.testing.assert_allclose(grads[k], v, rtol=0, atol=3e-5 * scale, err_msg=k)
