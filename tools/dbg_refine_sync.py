"""c4-like training with device-side densification for `--steps` iterations (refinement every 100), nothing else:
run under `rocprofv3 --hip-trace --stats` with two step counts -- the difference of the HIP API call counts is what
200 more iterations (two more refinements) cost in host-side synchronisation (tools/gpu_c4_r02.sh)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splat_one_amd.scene import pinhole_K, ring_cameras          # noqa: E402
from splat_one_amd.strategy import DefaultStrategy               # noqa: E402
from splat_one_amd.trainer import Config, Runner                 # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=2560)
ap.add_argument("--height", type=int, default=1440)
ap.add_argument("--strategy", default="default", choices=["default", "mcmc"])
a = ap.parse_args()
dev = torch.device("cuda:0")
W, H = a.width, a.height
strat = DefaultStrategy(refine_start_iter=0, refine_every=100, reset_every=3000, grow_grad2d=1.5e-4)
if a.strategy == "mcmc":         # the reference's `mcmc` preset (gsplat_trainer.py:975-983): relocate + add 5 % every 100 iterations, noise every iteration
    from splat_one_amd.strategy import MCMCStrategy
    strat = MCMCStrategy(refine_start_iter=0, refine_every=100, cap_max=4 * a.n)
cfg = Config(init_num_pts=a.n, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True, strategy=strat,
             max_gaussians=4 * a.n, **({"opacity_reg": 0.01, "scale_reg": 0.01} if a.strategy == "mcmc" else {}))
r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
ring = ring_cameras(8)
Ks = pinhole_K(W, H)[None].to(dev)
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
targets = [torch.stack([(xx + 0.1 * v) % 1.0, (yy + 0.07 * v) % 1.0, 0.5 * (xx + yy)], -1)[None].contiguous().to(dev) for v in range(8)]
cams = [ring[v:v + 1].contiguous().to(dev) for v in range(8)]
for i in range(a.steps):
    r.train_step(cams[i % 8], Ks, targets[i % 8])
torch.cuda.synchronize()
rep = r._engine.refine_report()
print(f"steps {a.steps} refinements {r._engine.refinements} N {rep['n_new']} void {r._engine.void_steps}")
