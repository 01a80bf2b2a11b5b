"""cProfile of the first densification step of the fused path."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
from splat_one_amd.strategy import DefaultStrategy
dev = torch.device("cuda:0")
N, W, H = (int(a) for a in (sys.argv[1:4] or (100000, 1920, 1080)))
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True,
             strategy=DefaultStrategy(refine_start_iter=20, refine_every=10))
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for i in range(30):
    r.train_step(c2w, Ks, pixels)
torch.cuda.synchronize()
for step in (30, 40):
    pr = cProfile.Profile(); pr.enable()
    t0 = time.time()
    r.train_step(c2w, Ks, pixels)      # step 30 / 40: refine
    torch.cuda.synchronize()
    pr.disable()
    print(f"step {r.step - 1}: {1e3 * (time.time() - t0):.1f} ms, N = {len(r.splats['means'])}")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    for i in range(9):
        t0 = time.time()
        r.train_step(c2w, Ks, pixels)
        torch.cuda.synchronize()
        if i < 3:
            print(f"  step {r.step - 1} (after the refine): {1e3 * (time.time() - t0):.1f} ms")
