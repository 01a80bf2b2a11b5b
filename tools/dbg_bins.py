"""Step time of the fused engine against the bin capacity (slots per tile) of the binned list layout."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.engine import FusedEngine
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, W, H = (int(a) for a in (sys.argv[1:4] or (100000, 1920, 1080)))
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for cap in (None, 256, 1024, 4096, 16384, 0):
    cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
    r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, lr_gamma_means=r.lr_gamma, binned=(cap != 0), bin_capacity=cap or None)
    for _ in range(30):
        eng.set_views(c2w, Ks, pixels, schedule=True); eng.step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(300):
        eng.set_views(c2w, Ks, pixels, schedule=True); eng.step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 300
    print(f"bin_capacity {cap}: in use {eng.bin_capacity}  fullest tile {eng._fullest_tile()}  {dt * 1e3:.3f} ms/step  {1 / dt:.0f} it/s  void {eng.void_steps}")
