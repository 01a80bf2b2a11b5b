"""Tile size 8 vs 16 in the fused engine (one wave per 8x8 tile: the binning pass does the per-quadrant culling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd import _lib
from splat_one_amd.engine import FusedEngine
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, W, H = (int(a) for a in (sys.argv[1:4] or (100000, 1920, 1080)))
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for ts in (16, 8):
    cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
    r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, tile_size=ts, lr_gamma_means=r.lr_gamma)
    for _ in range(30):
        eng.set_views(c2w, Ks, pixels, schedule=True); eng.step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(200):
        eng.set_views(c2w, Ks, pixels, schedule=True); eng.step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 200
    print(f"tile_size {ts}: {dt * 1e3:.3f} ms/step  {1 / dt:.0f} it/s  n_isects {eng.stats()['n_isects']}")
    eng.use_graph = False
    _lib.load().so_profile_enable(1)
    for _ in range(50):
        eng.set_views(c2w, Ks, pixels, schedule=True); eng.step()
    for k, (n, ms) in sorted(_lib.stage_profile().items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:22s} {ms * 1e3:8.1f} us")
    _lib.load().so_profile_enable(0)
