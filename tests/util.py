"""Shared helpers for the parity tests."""
import math

import numpy as np
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


# --------------------------------------------------------------- OpenSfM fixture writer (rows f3/f4)
def _shot(rotvec, center, camera):
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(rotvec).as_matrix()          # world->camera
    t = -R @ np.asarray(center, dtype=float)
    return {"rotation": list(map(float, rotvec)), "translation": list(map(float, t)), "camera": camera}


def write_opensfm_scene(tmp_path, n_shots=17, k1=0.0, k2=0.0, width=64, height=48, spherical=False, two_recs=False,
                         images=True, image_fn=None):
    import json
    import os
    rng = np.random.default_rng(5)
    cam_name = "v2 unknown unknown 64 48 perspective 0.9"
    cams = {cam_name: {"projection_type": "perspective", "width": width, "height": height, "focal": 0.9,
                       "k1": k1, "k2": k2}}
    if spherical:
        cams = {"pano": {"projection_type": "spherical", "width": width, "height": height}}
        cam_name = "pano"
    shots, centers = {}, []
    for i in range(n_shots):
        th = 2 * math.pi * i / n_shots
        c = np.array([4 * math.cos(th), 4 * math.sin(th), 0.3 * math.sin(3 * th)])
        rv = rng.normal(size=3) * 0.8
        shots[f"img_{i:03d}.png"] = _shot(rv, c, cam_name)
        centers.append(c)
    pts = {str(i): {"coordinates": list(map(float, rng.normal(size=3) * [2.0, 1.0, 0.2])),
                    "color": [int(v) for v in rng.integers(0, 256, 3)]} for i in range(200)}
    rec = {"cameras": cams, "shots": shots, "points": pts,
           "reference_lla": {"latitude": 35.0, "longitude": 139.0, "altitude": 10.0}}
    recs = [rec]
    if two_recs:
        rec2 = json.loads(json.dumps(rec))
        rec2["shots"] = {k.replace("img_", "b_"): v for k, v in rec2["shots"].items()}
        rec2["reference_lla"] = {"latitude": 35.0005, "longitude": 139.001, "altitude": 12.5}
        recs.append(rec2)
    with open(tmp_path / "reconstruction.json", "w") as f:
        json.dump(recs, f)
    if images:
        os.makedirs(tmp_path / "images", exist_ok=True)
        from PIL import Image as PILImage
        for name in list(shots) + (list(recs[1]["shots"]) if two_recs else []):
            arr = image_fn(name) if image_fn is not None else rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
            PILImage.fromarray(arr).save(tmp_path / "images" / name)
    return recs, np.array(centers)
