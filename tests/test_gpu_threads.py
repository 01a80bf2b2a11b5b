"""Two threads, as the reference runs them: the GUI thread renders the current camera while the training thread
steps (/root/reference/app/gsplat_manager.py:185 and the 1 Hz timer :151-165 against :204-206 `Thread(target=
runner.train)`), with no lock on the reference's side.  Here: `_viewer_render_fn` in a loop on thread B while thread A
runs fused training iterations with densification -- device-side refinements toggling the model set, hipGraph
captures and replays, lazy re-pointing of the torch-side handles.  No fault, no exception, finite frames."""
import threading

import numpy as np
import pytest
import torch

from splat_one_amd.scene import front_camera, pinhole_K, ring_cameras

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("device_refine", [True, False])
def test_viewer_thread_renders_while_training_with_densification(dev, device_refine):
    from splat_one_amd.strategy import DefaultStrategy
    from splat_one_amd.trainer import Config, Runner
    W, H, N = 160, 120, 4000
    strat = DefaultStrategy(refine_start_iter=4, refine_every=5, reset_every=20, refine_stop_iter=1000, grow_grad2d=5e-5)
    cfg = Config(init_num_pts=N, init_scale=0.3, init_opa=0.3, shN_init_std=0.05, sh_degree_interval=3, max_steps=200,
                 strategy=strat, fused=True, device_refine=device_refine)
    r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
    c2w = ring_cameras(4).to(dev)
    Ks = pinhole_K(W, H)[None].to(dev)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    target = torch.stack([xx, yy, 0.5 * (xx + yy)], -1)[None].to(dev).contiguous()
    r.train_step(c2w[0:1].contiguous(), Ks, target)           # the engine exists before the viewer starts
    stop, frames, errors, sizes = threading.Event(), [], [], set()
    view_c2w, view_K = front_camera(7.0).numpy(), pinhole_K(96, 64).numpy()

    def viewer():
        torch.cuda.set_device(0)
        try:
            while not stop.is_set():
                img = r._viewer_render_fn((view_c2w, view_K), (96, 64), camera_model="pinhole")
                frames.append((bool(np.isfinite(img).all()), float(img.max())))
                sizes.add(len(r.splats["means"]))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    th = threading.Thread(target=viewer, daemon=True)
    th.start()
    try:
        for step in range(1, 51):
            r.train_step(c2w[step % 4:step % 4 + 1].contiguous(), Ks, target)
        torch.cuda.synchronize()
    finally:
        stop.set()
        th.join(timeout=60)
    assert not th.is_alive() and not errors, errors
    assert len(frames) >= 3 and all(ok for ok, _ in frames) and max(m for _, m in frames) > 0.0
    assert r.step == 51 and len(sizes) >= 2                    # the viewer saw the set change under it
    n = len(r.splats["means"])
    for k, p in r.splats.items():
        assert p.shape[0] == n and torch.isfinite(p).all(), k
    st = r.optimizers["means"].state[r.splats["means"]]
    assert float(st["step"]) == 51.0 - r._engine.void_steps
