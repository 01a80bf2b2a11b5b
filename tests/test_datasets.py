"""Row f3 (dataset ingestion): host-side OpenSfM parser, normalisation and trajectories against
the golden vectors produced by the reference's own functions (tests/golden/make_golden.py) and
against hand-built reconstruction.json fixtures.  CPU only."""
import json
import math
import os

import numpy as np
import pytest
import torch

import importlib

from util import write_opensfm_scene as _write_scene

nz = importlib.import_module("splat_one_amd.datasets.normalize")
osfm = importlib.import_module("splat_one_amd.datasets.opensfm")
tj = importlib.import_module("splat_one_amd.datasets.traj")

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold(name):
    return np.load(os.path.join(GOLD, name))


# ------------------------------------------------------------------ golden: normalize.py
def test_normalize_matches_reference_golden():
    g = _gold("g3_normalize.npz")
    T1 = nz.similarity_from_cameras(g["c2w"])
    np.testing.assert_allclose(T1, g["T1"], rtol=0, atol=1e-12)
    c1 = nz.transform_cameras(T1, g["c2w"])
    np.testing.assert_allclose(c1, g["c1"], rtol=0, atol=1e-12)
    p1 = nz.transform_points(T1, g["pts"])
    np.testing.assert_allclose(p1, g["p1"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(nz.align_principle_axes(p1), g["T2"], rtol=0, atol=1e-10)
    cn, pn, T = nz.normalize(g["c2w"], g["pts"])
    np.testing.assert_allclose(cn, g["cn"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(pn, g["pn"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(T, g["T"], rtol=0, atol=1e-10)


def test_similarity_center_methods_and_errors():
    g = _gold("g3_normalize.npz")
    T = nz.similarity_from_cameras(g["c2w"], strict_scaling=True, center_method="poses")
    c = nz.transform_cameras(T, g["c2w"])
    # strict scaling: the farthest camera sits on the unit sphere; rotations stay orthonormal
    assert abs(np.linalg.norm(c[:, :3, 3], axis=1).max() - 1.0) < 1e-12
    np.testing.assert_allclose(c[:, :3, :3] @ c[:, :3, :3].transpose(0, 2, 1), np.tile(np.eye(3), (8, 1, 1)), atol=1e-12)
    with pytest.raises(ValueError):
        nz.similarity_from_cameras(g["c2w"], center_method="nope")


def test_similarity_y_up_input_uses_half_turn():
    # every camera upside-down (camera up = world +y... i.e. antiparallel to camera-space up):
    # the reference takes the fixed half-turn diag(-1,1,1) there (normalize.py:33-36)
    c2w = np.tile(np.eye(4), (4, 1, 1))
    c2w[:, :3, :3] = np.diag([-1.0, -1.0, 1.0])
    c2w[:, :3, 3] = np.array([[1, 0, 0], [0, 0, 1], [-1, 0, 0], [0, 0, -1]], dtype=float) * 3
    T = nz.similarity_from_cameras(c2w)
    R = T[:3, :3] / np.linalg.norm(T[0, :3])
    np.testing.assert_allclose(R, np.diag([-1.0, 1.0, 1.0]), atol=1e-15)


# ------------------------------------------------------------------ golden: traj.py
def test_viewmatrix_and_ellipse_z_golden():
    g = _gold("g4_traj.npz")
    vm = np.stack([tj.viewmatrix(g["look"][i], g["up"][i], g["pos"][i]) for i in range(4)])
    np.testing.assert_allclose(vm, g["viewmatrix"], atol=1e-14)
    ell = tj.generate_ellipse_path_z(g["ell_in"][:, :3, :], n_frames=8)
    np.testing.assert_allclose(ell, g["ell_out"], atol=1e-12)


def test_traj_paths_golden():
    g = _gold("g8_traj_paths.npz")
    poses, bounds = g["poses"], g["bounds"]
    np.testing.assert_allclose(tj.focus_point_fn(poses), g["focus"], atol=1e-12)
    np.testing.assert_allclose(tj.average_pose(poses), g["avg"], atol=1e-13)
    np.testing.assert_allclose(
        tj.generate_ellipse_path_y(poses, n_frames=12, variation=0.3, phase=0.25, height=0.5), g["ell_y"], atol=1e-12)
    np.testing.assert_allclose(
        tj.generate_ellipse_path_z(poses, n_frames=12, variation=0.2, phase=0.1, height=-0.3), g["ell_z"], atol=1e-12)
    np.testing.assert_allclose(tj.generate_spiral_path(poses, bounds, n_frames=16), g["spiral"], atol=1e-12)
    out = tj.generate_interpolated_path(poses, 3)
    assert out.shape == g["interp"].shape == (27, 3, 4)
    np.testing.assert_allclose(out, g["interp"], atol=1e-9)


# ------------------------------------------------------------------ golden: rotation helpers
def test_rotation_helpers_golden():
    g = _gold("g7_opensfm_math.npz")
    q = np.stack([osfm.angle_axis_to_quaternion(a) for a in g["angle_axis"]])
    np.testing.assert_allclose(q, g["qvec"], atol=1e-15)
    R = np.stack([osfm.qvec2rotmat(v) for v in q])
    np.testing.assert_allclose(R, g["R"], atol=1e-15)
    qb = np.stack([osfm.rotmat2qvec(m) for m in R])
    np.testing.assert_allclose(qb, g["qvec_back"], atol=1e-12)
    # and against the exponential map itself
    from scipy.spatial.transform import Rotation
    np.testing.assert_allclose(R, Rotation.from_rotvec(g["angle_axis"]).as_matrix(), atol=1e-12)
    np.testing.assert_array_equal(osfm.angle_axis_to_quaternion([0.0, 0.0, 0.0]), [1.0, 0.0, 0.0, 0.0])


# ------------------------------------------------------------------ UTM
def test_utm_forward_known_answers():
    assert osfm.utm_zone(2.2945) == 31 and osfm.utm_zone(-79.387139) == 17 and osfm.utm_zone(139.7) == 54
    e, n = osfm.utm_forward(-79.387139, 43.642566, 17)      # CN tower
    assert abs(e - 630084.0) < 1.5 and abs(n - 4833438.6) < 1.5
    # central meridian: false easting exactly, northing = k0 * meridian arc (WGS84 0->45 deg: 4984944.378 m)
    e, n = osfm.utm_forward(3.0, 45.0, 31)
    assert abs(e - 500000.0) < 1e-6 and abs(n - 0.9996 * 4984944.378) < 0.01
    e, n = osfm.utm_forward(4.0, 0.0, 31)
    assert abs(n) < 1e-6 and e > 500000
    # southern latitudes are negative northings (no false northing), mirror of the north
    e_s, n_s = osfm.utm_forward(4.0, -12.5, 31)
    e_n, n_n = osfm.utm_forward(4.0, 12.5, 31)
    assert abs(e_s - e_n) < 1e-6 and abs(n_s + n_n) < 1e-6


# ------------------------------------------------------------------ parser on a hand-built file
def test_parser_perspective_single_reconstruction(tmp_path):
    recs, centers = _write_scene(tmp_path)
    p = osfm.Parser(str(tmp_path), factor=1, normalize=False, test_every=8)
    assert len(p.images) == 17 and p.image_names[3] == "img_003.png"
    assert p.image_paths[3] == str(tmp_path) + "/images/img_003.png"
    # camera-to-world translation is the shot's optical centre; rotation is the transpose of the shot's
    np.testing.assert_allclose(p.camtoworlds[:, :3, 3], centers, atol=1e-12)
    from scipy.spatial.transform import Rotation
    for i, s in enumerate(recs[0]["shots"].values()):
        np.testing.assert_allclose(p.camtoworlds[i, :3, :3], Rotation.from_rotvec(s["rotation"]).as_matrix().T, atol=1e-12)
    # K: fx = fy = focal*width, principal point at the image centre (opensfm.py:425-429, :178-181)
    cid = p.camera_ids[0]
    assert cid == 1 and set(p.camera_ids) == {1}
    np.testing.assert_allclose(p.Ks_dict[cid], [[0.9 * 64, 0, 32], [0, 0.9 * 64, 24], [0, 0, 1]], atol=1e-9)
    np.testing.assert_array_equal(p.params_dict[cid], [0, 0, 0, 0])
    # zero distortion still goes through the OpenCV ROI logic: one pixel lost on each axis
    assert p.roi_undist_dict[cid] == [0, 0, 63, 47] and p.imsize_dict[cid] == (63, 47)
    assert p.points.dtype == np.float32 and p.points.shape == (200, 3)
    assert p.points_rgb.dtype == np.uint8 and p.points_err.shape == (200, 1)
    np.testing.assert_array_equal(p.transform, np.eye(4))
    c = centers - centers.mean(0)
    assert abs(p.scene_scale - np.linalg.norm(c, axis=1).max()) < 1e-12


def test_parser_factor_and_normalize(tmp_path):
    _write_scene(tmp_path)
    p0 = osfm.Parser(str(tmp_path), factor=1, normalize=False)
    p = osfm.Parser(str(tmp_path), factor=2, normalize=True)
    np.testing.assert_allclose(p.Ks_dict[1], [[0.9 * 32, 0, 16], [0, 0.9 * 32, 12], [0, 0, 1]], atol=1e-9)
    assert p.imsize_dict[1] == (31, 23)
    cn, pn, T = nz.normalize(p0.camtoworlds, p0.points)
    np.testing.assert_allclose(p.camtoworlds, cn, atol=1e-12)
    np.testing.assert_allclose(p.points, pn, atol=1e-6)
    np.testing.assert_allclose(p.transform, T, atol=1e-12)
    # normalised: median camera distance from the focus is ~1 and the thinnest cloud axis is z
    assert p.points[:, 2].std() < p.points[:, 1].std() < p.points[:, 0].std()


def test_dataset_split_items_and_lookup(tmp_path):
    _write_scene(tmp_path)
    p = osfm.Parser(str(tmp_path), factor=2, test_every=8)
    tr, va, al = osfm.Dataset(p, "train"), osfm.Dataset(p, "val"), osfm.Dataset(p, "all")
    assert list(va.indices) == [0, 8, 16] and len(tr) == 14 and len(al) == 17
    assert 0 not in tr.indices and 8 not in tr.indices
    item = tr[0]
    assert item["image_name"] == "img_001.png" and item["image_id"] == 0
    assert item["K"].dtype == torch.float32 and item["camtoworld"].shape == (4, 4)
    assert item["image"].shape == (23, 31, 3) and item["image"].dtype == torch.float32
    assert 0 <= float(item["image"].min()) and float(item["image"].max()) <= 255
    # factor 2 = exact 2x2 box average of the file, then the ROI crop
    from PIL import Image as PILImage
    raw = np.asarray(PILImage.open(tmp_path / "images" / "img_001.png")).astype(np.float64)
    box = raw.reshape(24, 2, 32, 2, 3).mean(axis=(1, 3))
    np.testing.assert_allclose(item["image"].numpy(), np.rint(box)[:23, :31], atol=1.0)
    assert tr.get_data_by_image_name("img_000.png") is None            # a val image
    np.testing.assert_array_equal(tr.get_data_by_image_name("img_002.png")["image"], tr[1]["image"])
    d = osfm.Dataset(p, "train", patch_size=8, load_depths=True)
    np.random.seed(0)
    it = d[2]
    assert it["image"].shape == (8, 8, 3) and it["depths"].shape == (8, 8)
    assert it["K"][0, 2] <= p.Ks_dict[1][0, 2] and it["K"][1, 2] <= p.Ks_dict[1][1, 2]


def test_parser_two_reconstructions_keep_all_shots_and_shift_points(tmp_path):
    recs, _ = _write_scene(tmp_path, two_recs=True, images=False)
    p = osfm.Parser(str(tmp_path))
    assert len(p.images) == 34 and p.image_names[17] == "b_000.png"
    assert set(p.camera_ids) == {1}                                     # same camera name -> same id
    off = p.images[17].diff_ref
    # 0.001 deg of longitude at 35N ~ 91.2 m, 0.0005 deg of latitude ~ 55.5 m; zone 54's grid north is
    # turned ~1.15 deg there (2 deg west of the central meridian), so compare the length and a loose box
    assert abs(math.hypot(off[0], off[1]) - math.hypot(91.2, 55.47)) < 0.2
    assert 90.0 < off[0] < 94.0 and 52.0 < off[1] < 57.0 and abs(off[2] - 2.5) < 1e-12
    np.testing.assert_array_equal(p.images[0].diff_ref, [0, 0, 0])
    a, b = p.points3D[:200], p.points3D[200:]
    np.testing.assert_allclose(b - a, np.tile([off[0], off[1], -off[2]], (200, 1)), atol=1e-9)


def test_parser_spherical_and_errors(tmp_path):
    _write_scene(tmp_path, spherical=True, images=False, width=2048, height=1024)
    p = osfm.Parser(str(tmp_path), factor=2)
    assert p.camera_ids[0] == 0 and p.cameras[0].panorama and p.camtype_dict[0] == "spherical"
    np.testing.assert_array_equal(p.Ks_dict[0], [[256, 0, 0], [0, 256, 0], [1024, 512, 1]])
    assert p.imsize_dict[0] == (1024, 512) and len(p.params_dict[0]) == 0 and 0 not in p.roi_undist_dict
    bad = {"cameras": {"c": {"projection_type": "brown", "width": 4, "height": 4}}, "shots": {}, "points": {}}
    (tmp_path / "reconstruction.json").write_text(json.dumps([bad]))
    with pytest.raises(ValueError, match="projection_type"):
        osfm.Parser(str(tmp_path))
    (tmp_path / "reconstruction.json").write_text("[]")
    with pytest.raises(ValueError):
        osfm.Parser(str(tmp_path))


# ------------------------------------------------------------------ undistortion / resampling
def test_undistort_maps_properties():
    K = np.array([[60.0, 0, 32], [0, 60.0, 24], [0, 0, 1]])
    Kn, roi, mx, my = osfm.undistort_maps(K, [0, 0, 0, 0], 64, 48)
    np.testing.assert_allclose(Kn, K, atol=1e-9)
    gx, gy = np.meshgrid(np.arange(64), np.arange(48), indexing="xy")
    np.testing.assert_allclose(mx, gx, atol=1e-4)
    np.testing.assert_allclose(my, gy, atol=1e-4)
    assert roi == [0, 0, 63, 47]
    # barrel distortion: undistorting pushes the border outwards, the inscribed rectangle is set by the
    # edge midpoints -> shorter focal, and every sample stays inside the file
    Kb, roi_b, mx, my = osfm.undistort_maps(K, [-0.2, 0.03, 0, 0], 64, 48)
    assert Kb[0, 0] < K[0, 0] and Kb[1, 1] < K[1, 1]
    assert mx.min() > -0.51 and mx.max() < 63.51 and my.min() > -0.51 and my.max() < 47.51
    assert roi_b == [0, 0, 63, 47]
    # the left/right edge midpoints of the new image sample the left/right edges of the file
    assert abs(mx[24, 0]) < 0.6 and abs(mx[24, 63] - 63) < 0.6
    # pincushion: the border moves inwards -> longer focal
    Kp, roi_p, mx, my = osfm.undistort_maps(K, [0.15, 0.0, 0, 0], 64, 48)
    assert Kp[0, 0] > K[0, 0] and roi_p[2] <= 64 and roi_p[3] <= 48 and np.isfinite(mx).all()
    with pytest.raises(NotImplementedError):
        osfm.undistort_maps(K, [0.1, 0, 0.01, 0], 64, 48)


def test_remap_and_resize():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (10, 12, 3), dtype=np.uint8)
    gx, gy = np.meshgrid(np.arange(12, dtype=np.float32), np.arange(10, dtype=np.float32), indexing="xy")
    np.testing.assert_array_equal(osfm.remap_bilinear(img, gx, gy), img)
    half = osfm.remap_bilinear(img.astype(np.float32), gx + 0.5, gy)
    np.testing.assert_allclose(half[:, :-1], 0.5 * (img[:, :-1].astype(np.float32) + img[:, 1:]), atol=1e-4)
    np.testing.assert_allclose(half[:, -1], 0.5 * img[:, -1], atol=1e-4)           # zero border
    assert osfm.remap_bilinear(img, gx - 5, gy - 20).max() == 0
    small = osfm.resize_area(img, (6, 5))
    np.testing.assert_array_equal(small, np.rint(img.reshape(5, 2, 6, 2, 3).astype(np.float64).mean(axis=(1, 3))).astype(np.uint8))
    # non-integer ratio: weights are fractional coverages and rows sum to one
    W = osfm._area_weights(10, 4)
    np.testing.assert_allclose(W.sum(1), 1.0)
    np.testing.assert_allclose(W[0, :3], [0.4, 0.4, 0.2])
    flat = osfm.resize_area(np.full((10, 7), 9, dtype=np.uint8), (3, 4))
    assert flat.shape == (4, 3) and (flat == 9).all()


def test_runner_wants_the_camera_model_of_the_shots(tmp_path):
    """360-degree shots train with Config.camera_model = "spherical" (the reference's default, gsplat_trainer.py:89);
    a perspective model on them is refused before anything touches the GPU."""
    from splat_one_amd.trainer import Config, Runner
    _write_scene(tmp_path, spherical=True, images=False, width=256, height=128)
    with pytest.raises(ValueError, match="spherical"):
        Runner.from_data_dir(0, 0, 1, Config(data_dir=str(tmp_path), data_factor=1, camera_model="pinhole"))
    with pytest.raises(ValueError, match="spherical"):          # the reference's own constructor call (:308-324)
        Runner(local_rank=0, world_rank=0, world_size=1, cfg=Config(data_dir=str(tmp_path), data_factor=1, camera_model="fisheye"))
    with pytest.raises(FileNotFoundError):
        Runner.from_data_dir(0, 0, 1, Config(data_dir=str(tmp_path / "nowhere")))
    # the Config surface of the reference (app/gsplat_manager.py:43-48 passes disable_viewer); branches outside the
    # rasterisation path are refused when switched on, before anything touches the GPU
    cfg = Config(data_dir=str(tmp_path), result_dir=str(tmp_path / "results"), disable_viewer=True, max_steps=30000)
    assert cfg.camera_model is None and cfg.port == 8080 and cfg.lpips_net == "alex" and cfg.depth_lambda == 1e-2
    for name in ("app_opt", "use_bilateral_grid"):
        with pytest.raises(NotImplementedError, match=name):
            Runner(0, 0, 1, Config(**{name: True}))
    with pytest.raises(NotImplementedError, match="compression"):
        Runner(0, 0, 1, Config(compression="png"))


# ------------------------------------------------------------------ camera_models.json (reference fixture)
def test_camera_models_merge_matches_reference_golden(tmp_path):
    """The reference's own CameraModelManager on the fixture of its tests/test_camera_models.py (overrides merged
    key-wise, unknown cameras added, default model on a missing / malformed file, merged table written back)."""
    cm = importlib.import_module("splat_one_amd.datasets.camera_models")
    gold = json.load(open(os.path.join(GOLD, "g9_camera_models.json")))
    for name, case in gold.items():
        d = tmp_path / name
        os.makedirs(d)
        if isinstance(case["base"], dict):
            json.dump(case["base"], open(d / "camera_models.json", "w"))
        elif isinstance(case["base"], str):
            (d / "camera_models.json").write_text(case["base"])
        if case["overrides"] is not None:
            json.dump(case["overrides"], open(d / "camera_models_overrides.json", "w"))
        merged = cm.load_camera_models(str(d))
        assert merged == case["merged"], name
        assert json.load(open(d / "camera_models.json")) == case["written_back"], name
    # the assertions of the reference's test_camera_model_manager_init, verbatim in meaning
    m = gold["override_one_field"]["merged"]
    assert m["Camera1"]["focal_ratio"] == 1.5 and m["Camera1"]["projection_type"] == "perspective"
    assert m["Camera2"]["projection_type"] == "spherical"
    K = cm.intrinsics(m["Camera1"])
    np.testing.assert_allclose(K.numpy(), [[1.5 * 1920, 0, 960], [0, 1.5 * 1920, 540], [0, 0, 1]])
    with pytest.raises(ValueError):
        cm.intrinsics(m["Camera2"])
