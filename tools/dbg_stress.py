"""Stress of the fused single-GPU path: many iterations over rotating views with densification (DefaultStrategy, then
MCMCStrategy): finite parameters, consistent optimiser state, how often buffers had to grow."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.trainer import Config, Runner
from splat_one_amd.strategy import DefaultStrategy, MCMCStrategy
dev = torch.device("cuda:0")
W, H, N = 640, 360, 50000
cams = ring_cameras(8).to(dev); Ks = pinhole_K(W, H)[None].to(dev)
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
targets = [torch.stack([(xx + 0.1 * v) % 1, yy, 0.5 * (xx + yy)], -1)[None].to(dev).contiguous() for v in range(8)]
for name, strat in (("default", DefaultStrategy(refine_start_iter=50, refine_every=50, reset_every=300, verbose=False)),
                    ("mcmc", MCMCStrategy(cap_max=120000, refine_start_iter=50, refine_every=50, verbose=False))):
    cfg = Config(init_num_pts=N, init_scale=0.3 if name == "default" else 0.1, init_opa=0.3, sh_degree_interval=100, fused=True, strategy=strat,
                 opacity_reg=0.01 if name == "mcmc" else 0.0, scale_reg=0.01 if name == "mcmc" else 0.0)
    r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        t0 = time.time()
        losses = []
        for step in range(700):
            v = (step * 3) % 8
            loss = r.train_step(cams[v:v + 1], Ks, targets[v])
            if step % 100 == 99:
                losses.append(float(loss))
        torch.cuda.synchronize()
        dt = time.time() - t0
    e = r._engine
    ok = all(torch.isfinite(p).all().item() for p in r.splats.values())
    st = r.optimizers["means"].state[r.splats["means"]]
    print(f"{name}: N {N} -> {len(r.splats['means'])}, finite {ok}, steps {r.step}, adam step {float(st['step'])} + void {e.void_steps}, "
          f"bins {e.bin_capacity}, fullest {e._fullest_tile()}, warnings {len(rec)}, {dt:.1f} s, losses {[round(l, 4) for l in losses]}")
    assert ok and float(st["step"]) + e.void_steps == 700
