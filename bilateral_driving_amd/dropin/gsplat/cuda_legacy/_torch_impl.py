"""gsplat.cuda_legacy._torch_impl.quat_to_rotmat (models/gaussians/basics.py:14; used by the
split sampling at vanilla.py:346 and rigid-node posing at nodes/rigid.py:339,407).  Pure torch,
differentiable, any device: it is host-side scene-graph plumbing, not part of the per-step kernels."""
import torch
import torch.nn.functional as F
from torch import Tensor


def quat_to_rotmat(quat: Tensor) -> Tensor:
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(F.normalize(quat, dim=-1), dim=-1)
    rows = (
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    )
    return torch.stack(rows, dim=-1).reshape(quat.shape[:-1] + (3, 3))
