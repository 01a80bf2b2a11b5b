"""Step-by-step comparison of Config.packed=True against packed=False through Runner.train_step (operator path)."""
import torch
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.strategy import DefaultStrategy
from splat_one_amd.trainer import Config, Runner

dev = torch.device("cuda:0")
W, H, B = 128, 96, 2
c2w = ring_cameras(8)[:B].to(dev)
Ks = pinhole_K(W, H)[None].repeat(B, 1, 1).to(dev)
pixels = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)


def mk(packed):
    strat = DefaultStrategy(refine_start_iter=2, refine_every=3, reset_every=50, refine_stop_iter=1000, grow_grad2d=5e-5,
                            refine_scale2d_stop_iter=100, verbose=True)
    cfg = Config(init_num_pts=3000, init_scale=0.3, init_opa=0.3, shN_init_std=0.05, sh_degree_interval=1, max_steps=200,
                 strategy=strat, fused=False, packed=packed, batch_size=B)
    r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
    with torch.no_grad():
        r.splats["scales"].add_((torch.randn(3000, 3, generator=torch.Generator().manual_seed(7)) * 0.4).to(dev))
    return r


a, b = mk(False), mk(True)
for step in range(5):
    la, lb = a.train_step(c2w, Ks, pixels), b.train_step(c2w, Ks, pixels)
    print(f"step {step}: loss {float(la):.7f} {float(lb):.7f}  N {len(a.splats['means'])} {len(b.splats['means'])}")
    if len(a.splats["means"]) != len(b.splats["means"]):
        break
    for k in a.splats.keys():
        d = (a.splats[k] - b.splats[k]).detach().abs()
        d = d.reshape(d.shape[0], -1).max(1).values
        print(f"   {k:10s} max {d.max().item():.3e}  rows > 1e-5: {(d > 1e-5).sum().item()}  first rows {torch.nonzero(d > 1e-5)[:6, 0].tolist()}")
    for k in ("grad2d", "count", "radii"):
        x, y = a.strategy_state[k], b.strategy_state[k]
        print(f"   state {k}: max diff {(x - y).abs().max().item():.3e}")
