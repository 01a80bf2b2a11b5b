import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd import _lib
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1/1.1)
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(230):
    r.train_step(c2w, Ks, pixels)
eng = r._engine; w = eng.ws
M = eng.M
blocks = 4 * M
stamps = torch.zeros(blocks, 6, dtype=torch.int64, device=dev)
vrec = torch.zeros_like(w["vrec"])
p = _lib.ptr
n_is = w["counters"][2 * M + 1:2 * M + 2]
torch.cuda.synchronize()
for variant in (0, 1, 4):
    for rep in range(2):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("so_debug_rasterize_bwd_wave_stamps", 1, N, W, H, p(w["rec"]), p(w["isect_offsets"]), p(w["flatten_ids"]), p(n_is),
                  p(w["render_alphas"]), p(w["last_ids"]), p(w["v_render_colors"]), p(w["zero_v_alphas"]), p(vrec), p(stamps), variant, _lib.stream())
        e1.record(); torch.cuda.synchronize()
    s = stamps.cpu().double()
    s = s[s[:, 2] > 0]
    t0, tl, te, npass, nvalid, nb = s.T
    dur = te - t0; pro = tl - t0; walk = te - tl
    print("variant %d (1=no atomic, 2=no reduction): kernel %.1f us | wave dur mean %.0f | prologue %.0f | walk %.0f | ticks/pass %.0f | passes/wave %.1f" %
          (variant, e0.elapsed_time(e1) * 1e3, dur.mean(), pro.mean(), walk.mean(), walk.sum() / npass.sum(), npass.mean()))
