"""Per-step wall time across the start of densification (default strategy: refine_start_iter 500, every 100)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
ts = []
for i in range(720):
    torch.cuda.synchronize(); t0 = time.time()
    r.train_step(c2w, Ks, pixels)
    torch.cuda.synchronize(); ts.append(time.time() - t0)
    if i in (499, 500, 501, 502, 599, 600, 601, 602, 700, 701):
        print(i, f"{ts[-1]*1e3:.2f} ms", "N =", len(r.splats["means"]), "capacity", r._engine.capacity, "void", r._engine.void_steps)
import statistics
for a, b in ((50, 450), (505, 595), (605, 695), (705, 719)):
    print(f"steps {a}-{b}: median {statistics.median(ts[a:b])*1e3:.3f} ms (synchronous per-step timing)")
