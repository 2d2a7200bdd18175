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


# ---- the slice folded in: features, grid-slot gradients and the guidance gradient as MFMA products -----------------------------------
def _region_case(levels, seed):
    """One 32-pixel tile inside one cell of every level: region node values [nch, gl, 2, 2], per-pixel cell data."""
    g = np.random.default_rng(seed)
    region = [(g.standard_normal((nch, gl, 2, 2)) * 0.5).astype(np.float32) for gl, nch in levels]
    gray = (g.random(32) * 1.2 - 0.1).astype(np.float32)
    cells = []
    for gl, nch in levels:
        iz = np.clip(gray * (gl - 1), 0, gl - 1).astype(np.float32)
        z0 = np.floor(iz).astype(np.int64); z1 = np.minimum(z0 + 1, gl - 1)
        v = gray * (gl - 1)
        cells.append(dict(fx=g.random(32, dtype=np.float32), fy=np.float32(g.random()), z0=z0, z1=z1, fz=(iz - z0).astype(np.float32),
                          interior=((v > 0) & (v < gl - 1)).astype(np.float32)))
    return region, cells, gray


def _torch_features(levels, region, cells, gray):
    """float64 trilinear sample of the region nodes at the tile's pixels, differentiable in the nodes and the gray value."""
    feats = []
    for (gl, nch), R, c in zip(levels, region, cells):
        iz = torch.clamp(gray * (gl - 1), 0, gl - 1)
        z0 = torch.tensor(c["z0"]); z1 = torch.tensor(c["z1"])
        fz = iz - z0
        fx = torch.tensor(c["fx"], dtype=torch.float64); fy = float(c["fy"])
        lo = R[:, z0]; hi = R[:, z1]                                # [nch, 32, 2, 2]
        plane = lambda P: ((P[..., 0, 0] * (1 - fx) + P[..., 0, 1] * fx) * (1 - fy) + (P[..., 1, 0] * (1 - fx) + P[..., 1, 1] * fx) * fy)
        feats.append((plane(lo) * (1 - fz) + plane(hi) * fz).T)     # [32, nch]
    return torch.cat(feats, dim=1)


@pytest.mark.parametrize("levels", [[(8, 24)], [(1, 8), (8, 8)], [(4, 16)], [(2, 8), (4, 8), (8, 8)]], ids=["single", "ms", "gl4", "three"])
def test_fused_slice_lane_model_equals_autograd(levels):
    region, cells, gray = _region_case(levels, seed=len(levels) * 10 + levels[0][0])
    F = sum(n for _, n in levels)
    g = np.random.default_rng(1)
    rgb = g.random((32, 3), dtype=np.float32)
    W1 = (g.standard_normal((64, F)) * 0.3).astype(np.float32)
    W2 = (g.standard_normal((64, 64)) * 0.2).astype(np.float32)
    W3 = (g.standard_normal((12, 64)) * 0.2).astype(np.float32)
    v_out = g.standard_normal((32, 3)).astype(np.float32)
    out, st = DM.fused_forward(levels, region, cells, rgb, W1, W2, W3)
    grads = DM.fused_backward(levels, region, cells, rgb, W1, W2, W3, v_out)
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    R = [t(r) for r in region]; gr = t(gray); c = t(rgb); w1, w2, w3 = t(W1), t(W2), t(W3)
    feats = _torch_features(levels, R, cells, gr)
    A = (torch.tanh(torch.tanh(feats @ w1.T) @ w2.T) @ w3.T).reshape(-1, 3, 4)
    ref = (A[..., :3] @ c[..., None])[..., 0] + A[..., 3] + c
    ref.backward(torch.tensor(v_out, dtype=torch.float64))
    np.testing.assert_allclose(DM.tile_to_matrix(st["xt"])[:F].T, feats.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    close = lambda a, b, what: np.testing.assert_allclose(a, b, rtol=0, atol=3e-5 * max(np.abs(b).max(), 1e-6), err_msg=what)
    close(grads["v_w1"], w1.grad.numpy(), "w1"); close(grads["v_w2"], w2.grad.numpy(), "w2"); close(grads["v_w3"], w3.grad.numpy(), "w3")
    close(grads["v_rgb"], c.grad.numpy(), "rgb (direct route)")
    close(grads["v_gray"], gr.grad.numpy(), "gray (guidance route)")
    for l, r in enumerate(R):
        close(grads["v_region"][l], r.grad.numpy(), f"grid nodes of level {l}")
