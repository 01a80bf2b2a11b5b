"""Host-overhead check of the sharded step: world=1 process group, exchange replaced by a device copy,
so the timing is kernels + Python/ctypes launch cost only (compare with the hipGraph'd FusedEngine)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
from splat_one_amd import sharded
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
sharded.all_to_all_rows = lambda out, inp, group=None: out.copy_(inp)
dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
r.sharded = True; r.world_size = 1
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(20):
    r.train_step(c2w, Ks, pixels)
torch.cuda.synchronize()
if len(sys.argv) > 1:      # "free": let the host run ahead of the GPU (no wait on the previous step's status event)
    r._engine._check_previous = lambda: None
for trial in range(2):
    t0 = time.time()
    for _ in range(200):
        r.train_step(c2w, Ks, pixels)
    t_host = time.time() - t0
    torch.cuda.synchronize()
    t_all = time.time() - t0
    print(f"sharded step (world=1, copy exchange): host issue {t_host / 200 * 1e3:.3f} ms/step, wall {t_all / 200 * 1e3:.3f} ms/step")
