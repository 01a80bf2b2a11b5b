"""Camera pose refinement (`Config.pose_opt`, `Config.pose_noise`): per-view deltas applied to camera-to-world
matrices before rendering; their gradient arrives through `viewmats = inv(camtoworlds)` and the projection
backward's v_viewmats (so_projection_bwd) plus the SH view directions.

Behaviour of the reference's utils/gsplat_utils/utils.py: `rotation_6d_to_matrix` (:117-138, Zhou et al. 2019,
Gram-Schmidt on two 3-vectors) and `CameraOptModule` (:12-49: an Embedding of 9 numbers per view = translation
delta + rotation delta added to the identity's 6D code; camtoworld' = camtoworld @ [R|t]).  Pinned by
tests/golden/g2_pose.npz (outputs of the reference module itself).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def rotation_6d_to_matrix(d6: Tensor) -> Tensor:
    """[..., 6] -> [..., 3, 3]: rows = orthonormalised first vector, second vector made orthogonal to it, their
    cross product."""
    u, v = d6[..., 0:3], d6[..., 3:6]
    r0 = F.normalize(u, dim=-1)
    r1 = F.normalize(v - (r0 * v).sum(dim=-1, keepdim=True) * r0, dim=-1)
    r2 = torch.linalg.cross(r0, r1, dim=-1)
    return torch.stack([r0, r1, r2], dim=-2)


class CameraOptModule(torch.nn.Module):
    """One learnable SE(3) delta per view."""

    def __init__(self, n: int):
        super().__init__()
        self.embeds = torch.nn.Embedding(n, 9)                     # [:3] translation, [3:] 6D rotation offset
        self.register_buffer("identity", torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0]))

    def zero_init(self) -> None:
        torch.nn.init.zeros_(self.embeds.weight)

    def random_init(self, std: float) -> None:
        torch.nn.init.normal_(self.embeds.weight, std=std)

    def forward(self, camtoworlds: Tensor, embed_ids: Tensor) -> Tensor:
        assert camtoworlds.shape[:-2] == embed_ids.shape, (camtoworlds.shape, embed_ids.shape)
        delta = self.embeds(embed_ids)                             # [..., 9]
        rot = rotation_6d_to_matrix(delta[..., 3:] + self.identity)
        top = torch.cat([rot, delta[..., :3, None]], dim=-1)       # [..., 3, 4]
        bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], device=delta.device, dtype=delta.dtype).expand(*top.shape[:-2], 1, 4)
        return camtoworlds @ torch.cat([top, bottom], dim=-2)
