"""Shared helpers for the parity tests."""
import math

import torch

from splat_one_amd.scene import lookat_c2w


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b|| / ||b||  (b = oracle), the per-tensor gradient metric of BASELINE.json north_star."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def small_scene(N=2000, seed=1, extent=3.0, scale=0.2, K=16, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    means = (torch.rand(N, 3, generator=g) * 2 - 1) * extent
    quats = torch.rand(N, 4, generator=g)
    scales = scale * (0.3 + torch.rand(N, 3, generator=g))
    opacities = torch.rand(N, generator=g) * 0.9 + 0.05
    sh = torch.randn(N, K, 3, generator=g) * 0.3
    return [t.to(dtype) for t in (means, quats, scales, opacities, sh)]


def two_cameras(W, H, fx=None):
    c2w = torch.stack([lookat_c2w((0.3, 0.2, -7.0)), lookat_c2w((4.0, 1.0, -5.0))])
    viewmats = torch.linalg.inv(c2w).contiguous()
    f = float(max(W, H)) if fx is None else fx
    Ks = torch.tensor([[[f, 0, W / 2.0], [0, f * 1.05, H / 2.0], [0, 0, 1]],
                       [[0.6 * f, 0, W * 0.48], [0, 0.6 * f, H * 0.52], [0, 0, 1]]], dtype=torch.float32)
    return viewmats, Ks
