"""Rows f3/f4 on the GPU: a Runner built from an OpenSfM directory trains through the HIP path,
writes/reads the reference's checkpoint layout, evaluates PSNR/SSIM (checked against the CPU
oracle's definitions), renders a trajectory and serves a viewer frame."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from util import write_opensfm_scene

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, **kw):
    from splat_one_amd.strategy import DefaultStrategy
    from splat_one_amd.trainer import Config
    base = dict(data_dir=str(tmp_path / "scene"), result_dir=str(tmp_path / "results"), data_factor=1,
                normalize_world_space=True, init_type="sfm", sh_degree=2, sh_degree_interval=5, max_steps=12,
                save_steps=[6], eval_steps=[6], strategy=DefaultStrategy(refine_start_iter=10_000))
    base.update(kw)
    return Config(**base)


def _scene(tmp_path, **kw):
    d = tmp_path / "scene"
    os.makedirs(d, exist_ok=True)
    yy, xx = np.meshgrid(np.linspace(0, 1, 48), np.linspace(0, 1, 64), indexing="ij")

    def image_fn(name):
        k = int(name.split("_")[1].split(".")[0])
        img = np.stack([xx, yy, 0.5 + 0.5 * np.sin(6 * xx + 0.3 * k)], -1)
        return (img * 255).astype(np.uint8)

    write_opensfm_scene(d, image_fn=image_fn, **kw)
    return d


@pytest.mark.parametrize("fused", [False, True])
def test_runner_from_data_dir_trains_saves_evals(dev, tmp_path, fused):
    from splat_one_amd.trainer import Runner
    _scene(tmp_path)
    cfg = _cfg(tmp_path, fused=fused)
    r = Runner.from_data_dir(0, 0, 1, cfg)
    assert len(r.trainset) == 14 and len(r.valset) == 3
    assert len(r.splats["means"]) == 200                                   # sfm init = the point cloud
    np.testing.assert_allclose(r.splats["means"].detach().cpu().numpy(), r.parser.points, atol=1e-6)
    assert abs(r.scene_scale - r.parser.scene_scale * 1.1) < 1e-9
    r.train()
    assert r.step == 12
    files = sorted(glob.glob(f"{cfg.result_dir}/ckpts/*.pt"))
    assert [os.path.basename(f) for f in files] == ["ckpt_11_rank0.pt", "ckpt_5_rank0.pt"]
    ck = torch.load(files[1], weights_only=True)
    assert ck["step"] == 5 and set(ck["splats"]) == {"means", "scales", "quats", "opacities", "sh0", "shN"}
    assert ck["splats"]["shN"].shape == (200, 8, 3)
    st = json.load(open(f"{cfg.result_dir}/stats/val_step0005.json"))
    assert set(st) == {"psnr", "ssim", "ellipse_time", "num_GS"} and st["num_GS"] == 200
    assert np.isfinite(st["psnr"]) and -1.0 <= st["ssim"] <= 1.0
    assert len(glob.glob(f"{cfg.result_dir}/renders/val_step5_*.png")) == 3
    tr = json.load(open(f"{cfg.result_dir}/stats/train_step0005_rank0.json"))
    assert set(tr) == {"mem", "ellipse_time", "num_GS"} and tr["num_GS"] == 200 and tr["mem"] > 0
    import yaml
    dumped = yaml.safe_load(open(f"{cfg.result_dir}/cfg.yml"))
    assert dumped["max_steps"] == 12 and dumped["data_factor"] == 1 and dumped["init_type"] == "sfm"


def test_eval_metrics_match_oracle_and_checkpoint_roundtrip(dev, tmp_path):
    from oracle import ssim_oracle
    from splat_one_amd.trainer import Runner
    _scene(tmp_path)
    cfg = _cfg(tmp_path)
    r = Runner.from_data_dir(0, 0, 1, cfg)
    r.train(4)
    stats = r.eval(step=3, save_images=False)
    # recompute both metrics on the CPU from the same renders, with the oracle's SSIM definition
    ps, ss = [], []
    for i in range(len(r.valset)):
        d = r.valset[i]
        px = d["image"][None].to(dev) / 255.0
        h, w = px.shape[1:3]
        with torch.no_grad():
            col, _, _ = r.rasterize_splats(d["camtoworld"][None].to(dev), d["K"][None].to(dev), w, h, sh_degree=cfg.sh_degree)
        col = col.clamp(0, 1).double().cpu()
        px = px.double().cpu()
        ps.append(10 * torch.log10(1.0 / ((col - px) ** 2).mean()))
        ss.append(ssim_oracle.fused_ssim(col.permute(0, 3, 1, 2), px.permute(0, 3, 1, 2), padding="valid"))
    assert abs(stats["psnr"] - torch.stack(ps).mean().item()) < 1e-3
    assert abs(stats["ssim"] - torch.stack(ss).mean().item()) < 1e-4

    path = r.save_checkpoint()
    assert path.endswith("ckpt_3_rank0.pt")
    r2 = Runner.from_data_dir(0, 0, 1, _cfg(tmp_path, init_type="random", init_num_pts=50))
    assert r2.load_checkpoints([path]) == 3 and r2.step == 4
    for k in r.splats.keys():
        assert torch.equal(r.splats[k].detach(), r2.splats[k].detach())
    stats2 = r2.eval(step=3, save_images=False)
    assert abs(stats2["psnr"] - stats["psnr"]) < 1e-5 and abs(stats2["ssim"] - stats["ssim"]) < 1e-6
    # two shards concatenate (the reference's multi-rank layout) and training resumes on them
    assert r2.load_checkpoints([path, path]) == 3 and len(r2.splats["means"]) == 400
    r2.train(2)
    assert r2.step == 6 and all(torch.isfinite(p).all() for p in r2.splats.values())
    with pytest.raises(ValueError):
        r2.load_checkpoints([])


def test_render_traj_and_viewer_frame(dev, tmp_path):
    from splat_one_amd.trainer import Runner
    _scene(tmp_path)
    cfg = _cfg(tmp_path, render_traj_path="interp")
    r = Runner.from_data_dir(0, 0, 1, cfg)
    path = r.trajectory()
    assert path.shape == (6, 4, 4)                                         # 17 shots - 10 trimmed = 7 keys -> 6
    np.testing.assert_allclose(path[:, 3], np.tile([0, 0, 0, 1.0], (6, 1)))
    frames = r.render_traj(step=0)
    assert frames.shape == (6, 47, 2 * 63, 3) and frames.dtype == np.uint8
    assert len(glob.glob(f"{cfg.result_dir}/videos/traj_0/*.png")) == 6
    cfg.render_traj_path = "ellipse"
    assert r.trajectory().shape == (120, 4, 4)
    cfg.render_traj_path = "spiral"
    with pytest.raises(ValueError):
        r.trajectory()
    cfg.render_traj_path = "zigzag"
    with pytest.raises(ValueError):
        r.trajectory()

    d = r.valset[0]
    img = r._viewer_render_fn((d["camtoworld"], d["K"]), (63, 47))
    assert img.shape == (47, 63, 3) and img.dtype == np.float32 and np.isfinite(img).all()
    with torch.no_grad():
        ref, _, _ = r.rasterize_splats(d["camtoworld"][None].to(dev), d["K"][None].to(dev), 63, 47,
                                       sh_degree=cfg.sh_degree, radius_clip=3.0)
    np.testing.assert_array_equal(img, ref[0].cpu().numpy())

    class State:                                                            # nerfview.CameraState surface
        c2w = d["camtoworld"].numpy()

        @staticmethod
        def get_K(wh):
            return d["K"].numpy()
    np.testing.assert_array_equal(r._viewer_render_fn(State(), (63, 47)), img)


def test_pose_opt_training_and_checkpoint(dev, tmp_path):
    """Config.pose_opt / pose_noise through Runner.train: the per-view deltas train (operator-level path), are
    decayed by the same schedule as the means, and are saved under the reference's checkpoint key."""
    from splat_one_amd.trainer import Runner
    _scene(tmp_path)
    cfg = _cfg(tmp_path, pose_opt=True, pose_noise=1e-3, pose_opt_lr=1e-3, fused=True, max_steps=6, save_steps=[6], eval_steps=[])
    r = Runner.from_data_dir(0, 0, 1, cfg)
    assert r.pose_adjust.embeds.weight.shape == (14, 9) and float(r.pose_adjust.embeds.weight.detach().abs().sum()) == 0.0
    r.train()
    w = r.pose_adjust.embeds.weight.detach()
    assert torch.isfinite(w).all() and float(w.abs().max()) > 0            # every visited view moved
    assert (w.abs().sum(1) > 0).sum().item() == 6                           # six steps, six distinct views
    ck = torch.load(glob.glob(f"{cfg.result_dir}/ckpts/ckpt_5_rank0.pt")[0], weights_only=True)
    assert "pose_adjust" in ck and torch.equal(ck["pose_adjust"]["embeds.weight"].to(w.device), w)
    lr = r.pose_optimizers[0].param_groups[0]["lr"]
    assert abs(lr - 1e-3 * (0.01 ** (6 / 6))) < 1e-9


@pytest.mark.parametrize("fused", [False, True])
def test_runner_trains_a_spherical_data_set(dev, tmp_path, fused):
    """The reference's default: 360-degree shots (`projection_type: spherical`, opensfm.py:176-193) rendered with
    Config.camera_model = "spherical" (gsplat_trainer.py:89).  Parser -> Dataset (no undistortion, the fork's K layout) ->
    training step -> eval, through both step implementations; the panorama sees the whole point cloud."""
    from splat_one_amd.trainer import Runner
    d = tmp_path / "scene"
    os.makedirs(d, exist_ok=True)
    Wp, Hp = 128, 64
    yy, xx = np.meshgrid(np.linspace(0, 1, Hp), np.linspace(0, 1, Wp), indexing="ij")

    def image_fn(name):
        k = int(name.split("_")[1].split(".")[0])
        return (np.stack([xx, yy, 0.5 + 0.5 * np.sin(6 * xx + 0.3 * k)], -1) * 255).astype(np.uint8)

    write_opensfm_scene(d, image_fn=image_fn, spherical=True, width=Wp, height=Hp)
    cfg = _cfg(tmp_path, fused=fused, disable_viewer=True)         # camera_model left at its default, as the GUI does
    r = Runner(local_rank=0, world_rank=0, world_size=1, cfg=cfg)  # app/gsplat_manager.py:43-49
    assert cfg.camera_model == "spherical" and {r.parser.camtype_dict[c] for c in r.parser.camera_ids} == {"spherical"}
    assert r.allset.get_data_by_image_name("img_003.png")["image"].shape == (Hp, Wp, 3)
    r.train()
    assert r.step == 12
    info = r.last_info
    radii = info["radii"] if "radii" in info else info["engine"].ws["radii"]
    assert int((radii > 0).sum()) >= 190                      # 200 SfM points all around the camera: (nearly) all visible
    st = json.load(open(f"{cfg.result_dir}/stats/val_step0005.json"))
    assert np.isfinite(st["psnr"]) and st["num_GS"] == 200
    frame = r._viewer_render_fn((r.parser.camtoworlds[0], np.eye(3)), (Wp, Hp), camera_model="spherical")
    assert frame.shape == (Hp, Wp, 3) and np.isfinite(frame).all() and frame.max() > 0
    with pytest.raises(ValueError, match="spherical"):
        Runner.from_data_dir(0, 0, 1, _cfg(tmp_path, camera_model="pinhole"))
