"""Distribution of the per-tile list lengths of a (possibly gathered) cloud: what the sort kernels are handed.
usage: dbg_listlens.py N cloud_scale"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.engine import FusedEngine
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, scale = int(sys.argv[1]), float(sys.argv[2])
W, H = 1920, 1080
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
with torch.no_grad():
    r.splats["means"].mul_(scale)
eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, lr_gamma_means=r.lr_gamma)
for _ in range(3):
    eng.set_views(c2w, Ks, pixels, schedule=True); eng.step()
torch.cuda.synchronize()
c = eng.ws["counters"][:eng.M].clamp(max=eng.bin_capacity).cpu()
print("bins", eng.bin_capacity, "tiles", eng.M, "entries", int(c.sum()), "fullest", int(c.max()))
for lo, hi in ((1, 256), (257, 2048), (2049, 4096), (4097, 8192), (8193, 16384), (16385, 32768), (32769, 1 << 30)):
    m = (c >= lo) & (c <= hi)
    print(f"{lo:6d}..{hi:<10d} tiles {int(m.sum()):5d}  keys {int(c[m].sum()):9d}")
