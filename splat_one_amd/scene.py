"""Synthetic scenes and cameras for tests and bench.py.

The splat initialisation restates the reference's own random init
(/root/reference/utils/gsplat_utils/gsplat_trainer.py:224-257, seed per :290 and
utils/gsplat_utils/utils.py:141-156); tests/test_golden.py pins it against fixtures produced by
running the reference's function.  Host-side, one-off: not part of the timed path.
"""
from __future__ import annotations

import math
import random
from typing import Dict, Tuple

import numpy as np
import torch
from torch import Tensor

SH_C0 = 0.28209479177387814


def set_random_seed(seed: int) -> None:
    """utils/gsplat_utils/utils.py:153-156"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def knn(x: Tensor, K: int = 4) -> Tensor:
    """Distances to the K nearest neighbours (self included), utils/gsplat_utils/utils.py:141-145."""
    from sklearn.neighbors import NearestNeighbors
    x_np = x.cpu().numpy()
    model = NearestNeighbors(n_neighbors=K, metric="euclidean").fit(x_np)
    distances, _ = model.kneighbors(x_np)
    return torch.from_numpy(distances).to(x)


def rgb_to_sh(rgb: Tensor) -> Tensor:
    """utils/gsplat_utils/utils.py:148-150"""
    return (rgb - 0.5) / SH_C0


def random_splats(n: int, init_extent: float = 3.0, init_opacity: float = 0.1, init_scale: float = 1.0,
                  scene_scale: float = 1.0, sh_degree: int = 3, world_rank: int = 0, world_size: int = 1,
                  shN_std: float = 0.0, shN_seed: int = 43) -> Dict[str, Tensor]:
    """init_type="random" branch of create_splats_with_optimizers (gsplat_trainer.py:224-257).
    Call set_random_seed(42 + rank) first to reproduce the reference's stream.  Returns CPU
    tensors in the reference's parameterisation: means, scales (log), quats (raw), opacities
    (logit), sh0 [N,1,3], shN [N,(deg+1)^2-1,3].  `shN_std>0` replaces the reference's zero
    higher-order coefficients by N(0,std) noise (bench/test scenes: every SH band gets gradient)."""
    points = init_extent * scene_scale * (torch.rand((n, 3)) * 2 - 1)
    rgbs = torch.rand((n, 3))
    dist2_avg = (knn(points, 4)[:, 1:] ** 2).mean(dim=-1)
    dist_avg = torch.sqrt(dist2_avg)
    scales = torch.log(dist_avg * init_scale).unsqueeze(-1).repeat(1, 3)
    points = points[world_rank::world_size]
    rgbs = rgbs[world_rank::world_size]
    scales = scales[world_rank::world_size]
    N = points.shape[0]
    quats = torch.rand((N, 4))
    opacities = torch.logit(torch.full((N,), init_opacity))
    colors = torch.zeros((N, (sh_degree + 1) ** 2, 3))
    colors[:, 0, :] = rgb_to_sh(rgbs)
    if shN_std > 0:
        g = torch.Generator().manual_seed(shN_seed)
        colors[:, 1:, :] = torch.randn(colors[:, 1:, :].shape, generator=g) * shN_std
    return {"means": points, "scales": scales, "quats": quats, "opacities": opacities,
            "sh0": colors[:, :1, :].contiguous(), "shN": colors[:, 1:, :].contiguous()}


def pinhole_K(width: int, height: int, focal_ratio: float = 1.0) -> Tensor:
    """fx = fy = focal_ratio * max(W,H) (app/camera_models.py:207-213, :230-237), principal point at
    the image centre (utils/datasets/opensfm.py:176-185)."""
    f = focal_ratio * max(width, height)
    return torch.tensor([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])


def lookat_c2w(position, target=(0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0)) -> Tensor:
    """OpenCV camera-to-world (+z forward, +y down) looking from `position` to `target`."""
    pos = torch.tensor(position, dtype=torch.float64)
    z = torch.tensor(target, dtype=torch.float64) - pos
    z = z / z.norm()
    upv = torch.tensor(up, dtype=torch.float64)
    x = torch.linalg.cross(z, upv)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    m = torch.eye(4, dtype=torch.float64)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, pos
    return m.to(torch.float32)


def front_camera(distance: float = 9.0) -> Tensor:
    """Identity rotation, position (0,0,-distance): looks down +z at the origin."""
    m = torch.eye(4)
    m[2, 3] = -distance
    return m


def ring_cameras(n: int = 8, radius: float = 9.0, height: float = 0.0) -> Tensor:
    """n cameras on a circle in the xz-plane looking at the origin; camera 0 is
    `front_camera`.  Matches tests/golden/g4_traj.npz 'ring' (built with the reference's viewmatrix)."""
    out = []
    for k in range(n):
        th = 2 * math.pi * k / n
        out.append(lookat_c2w((radius * math.sin(th), height, -radius * math.cos(th))))
    return torch.stack(out)


def make_scene(n: int, width: int, height: int, regime: str = "mcmc", n_views: int = 1, seed: int = 42,
               shN_std: float = 0.1) -> Tuple[Dict[str, Tensor], Tensor, Tensor]:
    """(splats, camtoworlds[C,4,4], Ks[C,3,3]) for the BASELINE configs.  regime 'ref' =
    init_scale 1.0 / init_opa 0.1 (trainer `default` preset, :117-119); 'mcmc' = 0.1 / 0.5 (:977-983)."""
    assert regime in ("ref", "mcmc"), regime
    set_random_seed(seed)
    init_scale, init_opa = (1.0, 0.1) if regime == "ref" else (0.1, 0.5)
    splats = random_splats(n, init_opacity=init_opa, init_scale=init_scale, shN_std=shN_std)
    c2w = front_camera()[None] if n_views == 1 else ring_cameras(n_views)
    Ks = pinhole_K(width, height)[None].repeat(c2w.shape[0], 1, 1)
    return splats, c2w, Ks
