"""Host time of one operator-path iteration, segment by segment (perf_counter, no profiler), 8 views cycled like bench.py."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.trainer import Config, Runner
from splat_one_amd.losses import photometric_loss
from splat_one_amd.optimizers import step_all
dev = torch.device("cuda:0")
N, W, H, NV = 100000, 1920, 1080, int(os.environ.get("NV", "8"))
from splat_one_amd.strategy import DefaultStrategy
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=False,
             raw_params_call=("--gsplat-call" not in sys.argv), strategy=DefaultStrategy(refine_start_iter=10**9, verbose=False))
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
ring = ring_cameras(8)
Ks = pinhole_K(W, H)[None].to(dev)
views = [(ring[v:v + 1].contiguous().to(dev), Ks, torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(v)).to(dev)) for v in range(NV)]
for i in range(40):
    r.train_step(*views[i % NV])
torch.cuda.synchronize()
for trial in range(5):
    t0 = time.perf_counter()
    for i in range(200):
        r.train_step(*views[i % NV])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        r.train_step(*views[i % NV])
    e1.record()
    torch.cuda.synchronize()
    print(f"train_step: host issue {t_host / 200 * 1e3:.3f} ms/step, wall {t_all / 200 * 1e3:.3f} ms/step; events {e0.elapsed_time(e1) / 100:.3f} ms/step", flush=True)
# the same iteration spelled out, timed per segment
seg = {}
def tick(name, t):
    seg[name] = seg.get(name, 0.0) + (time.perf_counter() - t)
    return time.perf_counter()
s = cfg.strategy
n_it = 200
torch.cuda.synchronize()
T0 = time.perf_counter()
for i in range(n_it):
    c2w, K, px = views[i % NV]
    t = time.perf_counter()
    renders, alphas, info = r.rasterize_splats(camtoworlds=c2w, Ks=K, width=W, height=H, sh_degree=3, near_plane=0.01, far_plane=1e8, render_mode="RGB")
    t = tick("rasterize_splats", t)
    s.step_pre_backward(params=r.splats, optimizers=r.optimizers, state=r.strategy_state, step=r.step, info=info)
    t = tick("pre_backward", t)
    loss, _, _ = photometric_loss(renders[..., 0:3], px, 0.2)
    t = tick("loss", t)
    loss.backward()
    t = tick("backward", t)
    step_all(r.optimizers.values(), set_to_none=True)
    t = tick("step_all", t)
    s.step_post_backward(params=r.splats, optimizers=r.optimizers, state=r.strategy_state, step=10, info=info, packed=False)
    t = tick("post_backward", t)
host = time.perf_counter() - T0
torch.cuda.synchronize()
print(f"spelled out: host {host / n_it * 1e6:.0f} us/step, wall {(time.perf_counter() - T0) / n_it * 1e6:.0f} us/step")
for k, v in seg.items():
    print(f"  {k:18s} {v / n_it * 1e6:7.1f} us")
