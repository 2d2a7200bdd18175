"""`nvdiffrast.torch.texture` for the reference's sky model (/root/reference/project/models/modules.py:202):
    dr.texture(base[None, ...], l, filter_mode='linear', boundary_mode='cube')
with base [6,res,res,C] and l [B,H,W,3].  Only this mode exists here (anything else raises); see
bilateral_driving_amd/csrc/envlight.hip for the convention and its parity status."""
from bilateral_driving_amd.envlight import cubemap_sample


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
    if boundary_mode != "cube" or filter_mode not in ("linear", "auto") or uv_da is not None or mip is not None or mip_level_bias is not None:
        raise NotImplementedError("only texture(tex, dirs, filter_mode='linear', boundary_mode='cube') is provided")
    if tex.dim() != 5 or tex.shape[1] != 6 or uv.shape[-1] != 3:
        raise ValueError("cube texture is [minibatch, 6, res, res, C] and uv is [minibatch, H, W, 3]")
    if tex.shape[0] != 1 and tex.shape[0] != uv.shape[0]:
        raise ValueError("minibatch mismatch")
    if tex.shape[0] == 1:
        return cubemap_sample(tex[0], uv)
    import torch
    return torch.stack([cubemap_sample(tex[b], uv[b]) for b in range(tex.shape[0])], dim=0)
