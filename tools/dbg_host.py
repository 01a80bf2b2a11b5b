"""Host issue time vs wall time per training step of the fused single-GPU path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=("--operator" not in sys.argv))
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(30):
    r.train_step(c2w, Ks, pixels)
torch.cuda.synchronize()
import cProfile, pstats
for trial in range(3):
    t0 = time.time()
    for _ in range(200):
        r.train_step(c2w, Ks, pixels)
    t_host = time.time() - t0
    torch.cuda.synchronize()
    t_all = time.time() - t0
    print(f"host issue {t_host / 200 * 1e3:.3f} ms/step, wall {t_all / 200 * 1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    r.train_step(c2w, Ks, pixels)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45 if "--operator" in sys.argv else 14)
