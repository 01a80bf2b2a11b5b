"""Generate tests/golden/*.npz by RUNNING the reference's own importable code.

Run only in the build container (needs /root/reference; that tree never travels to the GPU
box).  This script contains no reference source: importable modules are imported, and the one
function that lives in a module which cannot be imported here
(`create_splats_with_optimizers`, utils/gsplat_utils/gsplat_trainer.py:204-281 -- the module
imports imageio/tyro/gsplat/... at top level) is located in the reference file with `ast` at
run time and executed as-is with its three external names bound to the importable originals
(`knn`, `rgb_to_sh` from utils/gsplat_utils/utils.py) or to a sentinel (`SelectiveAdam`,
`Parser`: unused on the random-init / torch.optim.Adam path).

    python tests/golden/make_golden.py          # rewrites the fixtures next to this file

Fixtures (data only):
  g1_init_*.npz    splat init + optimiser hyper-parameters      (gsplat_trainer.py:204-281, :290)
  g2_pose.npz      rotation_6d_to_matrix / CameraOptModule       (utils.py:12-49, 117-138)
  g3_normalize.npz similarity_from_cameras/align_principle_axes  (utils/datasets/normalize.py)
  g4_traj.npz      viewmatrix / generate_ellipse_path_z          (utils/datasets/traj.py:16-142)
  g6_knn_sh.npz    knn, rgb_to_sh                                (utils.py:141-150)
  g7_opensfm_math.npz angle_axis_to_quaternion / qvec2rotmat / rotmat2qvec, extracted like g1 (the
                   module imports pyproj/cv2/imageio at top level)   (utils/datasets/opensfm.py:47-84)
  g9_camera_models.json CameraModelManager.load_camera_models (class extracted like g1: the module imports PyQt5)
                   on the fixture data of the reference's own tests/test_camera_models.py:13-41 (app/camera_models.py:225-292)
  g8_traj_paths.npz generate_ellipse_path_y / generate_interpolated_path / generate_spiral_path /
                   focus_point_fn / average_pose                 (utils/datasets/traj.py:25-255)
"""
import ast
import math
import os
import sys
from typing import Dict, Optional, Tuple

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    sys.path.insert(0, REF)
    import importlib
    U = importlib.import_module("utils.gsplat_utils.utils")
    Nz = importlib.import_module("utils.datasets.normalize")
    Tj = importlib.import_module("utils.datasets.traj")
    return U, Nz, Tj


def _extract_function(path: str, name: str, namespace: dict):
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), namespace)
            return namespace[name]
    raise KeyError(name)


def g1(U):
    ns = dict(torch=torch, math=math, knn=U.knn, rgb_to_sh=U.rgb_to_sh, Optional=Optional, Tuple=Tuple,
              Dict=Dict, Parser=object, SelectiveAdam=None)
    create = _extract_function(os.path.join(REF, "utils/gsplat_utils/gsplat_trainer.py"),
                               "create_splats_with_optimizers", ns)
    cases = [
        # name, N, init_opa, init_scale, batch_size, world_rank, world_size   (presets: trainer :117-119, :977-983)
        ("n256_ref", 256, 0.1, 1.0, 1, 0, 1),
        ("n256_mcmc", 256, 0.5, 0.1, 1, 0, 1),
        ("n256_bs8_rank1of2", 256, 0.1, 1.0, 4, 1, 2),
        ("n10k_ref", 10_000, 0.1, 1.0, 1, 0, 1),
    ]
    for name, N, opa, sc, bs, rank, world in cases:
        U.set_random_seed(42 + rank)                                  # gsplat_trainer.py:290
        splats, opts = create(None, init_type="random", init_num_pts=N, init_extent=3.0,
                              init_opacity=opa, init_scale=sc, scene_scale=1.0, sh_degree=3,
                              batch_size=bs, device="cpu", world_rank=rank, world_size=world)
        out = {}
        for k, v in splats.items():
            a = v.detach().numpy()
            if N <= 256:
                out[k] = a
            else:                                                     # keep the big case small
                out[k + "_head"] = a[:8]
                out[k + "_tail"] = a[-8:]
                out[k + "_sum"] = np.array(a.astype(np.float64).sum())
                out[k + "_abssum"] = np.array(np.abs(a.astype(np.float64)).sum())
                out[k + "_shape"] = np.array(a.shape)
        for k, o in opts.items():
            g = o.param_groups[0]
            out["opt_" + k] = np.array([g["lr"], g["eps"], g["betas"][0], g["betas"][1]], dtype=np.float64)
        out["args"] = np.array([N, opa, sc, bs, rank, world], dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, f"g1_init_{name}.npz"), **out)


def g2(U):
    g = torch.Generator().manual_seed(7)
    d6 = torch.randn(4, 6, generator=g)
    R = U.rotation_6d_to_matrix(d6)
    mod = U.CameraOptModule(5)
    with torch.no_grad():
        mod.embeds.weight.copy_(torch.randn(5, 9, generator=g) * 0.05)
    c2w = torch.eye(4).repeat(3, 1, 1)
    c2w[:, :3, 3] = torch.randn(3, 3, generator=g)
    c2w[:, :3, :3] = U.rotation_6d_to_matrix(torch.randn(3, 6, generator=g))
    ids = torch.tensor([4, 0, 2])
    out = mod(c2w, ids)
    np.savez_compressed(os.path.join(HERE, "g2_pose.npz"), d6=d6.numpy(), R=R.numpy(),
                        embeds=mod.embeds.weight.detach().numpy(), c2w=c2w.numpy(), ids=ids.numpy(),
                        out=out.detach().numpy())


def ring_cameras(Tj, n=8, radius=9.0, height=0.0):
    """n cameras on a circle in the xz-plane looking at the origin (OpenCV: +z forward, +y down)."""
    poses = []
    for k in range(n):
        th = 2 * math.pi * k / n
        pos = np.array([radius * math.sin(th), height, -radius * math.cos(th)])
        m = Tj.viewmatrix(-pos, np.array([0.0, -1.0, 0.0]), pos)      # [3,4]
        m = np.concatenate([m, np.array([[0, 0, 0, 1.0]])], 0)
        # viewmatrix builds (x,y,z) with y = "up"; OpenCV has y down -> flip x,y to keep right-handed
        m[:3, 0] *= -1
        m[:3, 1] *= -1
        poses.append(m)
    return np.stack(poses)


def g3(Nz, Tj):
    rng = np.random.default_rng(3)
    c2w = ring_cameras(Tj, 8, 9.0, 0.7)
    c2w[:, :3, 3] += rng.normal(0, 0.2, (8, 3))
    pts = rng.normal(0, 1.0, (1000, 3)) * np.array([3.0, 1.0, 2.0])
    T1 = Nz.similarity_from_cameras(c2w)
    c1 = Nz.transform_cameras(T1, c2w)
    p1 = Nz.transform_points(T1, pts)
    T2 = Nz.align_principle_axes(p1)
    cn, pn, T = Nz.normalize(c2w, pts)
    np.savez_compressed(os.path.join(HERE, "g3_normalize.npz"), c2w=c2w, pts=pts, T1=T1, c1=c1, p1=p1,
                        T2=T2, cn=cn, pn=pn, T=T)


def g4(Tj):
    rng = np.random.default_rng(4)
    look = rng.normal(size=(4, 3))
    up = rng.normal(size=(4, 3))
    pos = rng.normal(size=(4, 3))
    vm = np.stack([Tj.viewmatrix(look[i], up[i], pos[i]) for i in range(4)])
    ring = ring_cameras(Tj, 8, 9.0, 0.0)
    # the ellipse path needs z-up input poses (multinerf convention): rotate the ring so its plane is xy
    Rzx = np.array([[1.0, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]])
    poses = np.einsum("ij,njk->nik", Rzx, ring)
    ell = Tj.generate_ellipse_path_z(poses[:, :3, :], n_frames=8)
    np.savez_compressed(os.path.join(HERE, "g4_traj.npz"), look=look, up=up, pos=pos, viewmatrix=vm,
                        ring=ring, ell_in=poses, ell_out=ell)


def g6(U):
    g = torch.Generator().manual_seed(11)
    pts = torch.rand(500, 3, generator=g) * 6 - 3
    d = U.knn(pts, 4)
    rgb = torch.rand(64, 3, generator=g)
    np.savez_compressed(os.path.join(HERE, "g6_knn_sh.npz"), pts=pts.numpy(), knn4=d.numpy(),
                        rgb=rgb.numpy(), sh=U.rgb_to_sh(rgb).numpy())


def g7():
    path = os.path.join(REF, "utils/datasets/opensfm.py")
    ns = dict(np=np, math=math)
    aa2q = _extract_function(path, "angle_axis_to_quaternion", ns)
    q2r = _extract_function(path, "qvec2rotmat", ns)
    r2q = _extract_function(path, "rotmat2qvec", ns)
    rng = np.random.default_rng(7)
    aa = rng.normal(size=(16, 3)) * np.array([0.01, 0.3, 1.0, 2.5] * 4)[:, None]
    q = np.stack([aa2q(a) for a in aa])
    R = np.stack([q2r(v) for v in q])
    qb = np.stack([r2q(m) for m in R])
    np.savez_compressed(os.path.join(HERE, "g7_opensfm_math.npz"), angle_axis=aa, qvec=q, R=R, qvec_back=qb)


def g8(Tj):
    rng = np.random.default_rng(8)
    ring = ring_cameras(Tj, 10, 6.0, 0.4)
    ring[:, :3, 3] += rng.normal(0, 0.3, (10, 3))
    poses = ring[:, :3, :]
    bounds = np.array([0.5, 30.0])
    np.savez_compressed(
        os.path.join(HERE, "g8_traj_paths.npz"), poses=poses, bounds=bounds,
        focus=Tj.focus_point_fn(poses), avg=Tj.average_pose(poses),
        ell_y=Tj.generate_ellipse_path_y(poses, n_frames=12, variation=0.3, phase=0.25, height=0.5),
        ell_z=Tj.generate_ellipse_path_z(poses, n_frames=12, variation=0.2, phase=0.1, height=-0.3),
        interp=Tj.generate_interpolated_path(poses, 3),
        spiral=Tj.generate_spiral_path(poses, bounds, n_frames=16))


def _extract_class(path: str, name: str, namespace: dict):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == name:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
            return namespace[name]
    raise KeyError(name)


def g9():
    import json
    import tempfile
    Manager = _extract_class(os.path.join(REF, "app/camera_models.py"), "CameraModelManager", dict(os=os, json=json))
    base = {"Camera1": {"projection_type": "perspective", "width": 1920, "height": 1080, "focal_ratio": 1.2},
            "Camera2": {"projection_type": "spherical", "width": 3840, "height": 2160, "focal_ratio": 1.0}}
    cases = {"override_one_field": (base, {"Camera1": {"focal_ratio": 1.5}}),
             "override_adds_camera": (base, {"Camera3": {"projection_type": "fisheye", "width": 640, "height": 480, "focal_ratio": 0.8}}),
             "no_overrides": (base, None),
             "missing_base_file": (None, {"Perspective": {"focal_ratio": 0.9}}),
             "malformed_base_file": ("{not json", None)}
    out = {}
    for name, (b, o) in cases.items():
        with tempfile.TemporaryDirectory() as d:
            if isinstance(b, dict):
                json.dump(b, open(os.path.join(d, "camera_models.json"), "w"))
            elif isinstance(b, str):
                open(os.path.join(d, "camera_models.json"), "w").write(b)
            if o is not None:
                json.dump(o, open(os.path.join(d, "camera_models_overrides.json"), "w"))
            m = Manager(d)
            written = json.load(open(os.path.join(d, "camera_models.json")))
            out[name] = {"base": b, "overrides": o, "merged": m.get_camera_models(), "written_back": written}
    json.dump(out, open(os.path.join(HERE, "g9_camera_models.json"), "w"), indent=1)


if __name__ == "__main__":
    U, Nz, Tj = _load_reference()
    g1(U)
    g2(U)
    g3(Nz, Tj)
    g4(Tj)
    g6(U)
    g7()
    g8(Tj)
    g9()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
