"""Warm-loop timing of the per-Gaussian kernels (so_preprocess_fwd / _bwd) vs their in-sequence time:
tells whether they are slow because of cold instruction cache / launch ramp or intrinsically."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd import _lib
from splat_one_amd.engine import FusedEngine
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, W, H = (int(a) for a in (sys.argv[1:4] or (100000, 1920, 1080)))
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(10):
    r.train_step(c2w, Ks, pixels)
eng = r._engine; w = eng.ws; s = eng.splats; p = _lib.ptr; g = w["grads"]
M = eng.M
VARIANT = sys.argv[4] if len(sys.argv) > 4 else ""      # "" | nohist | nocull | deg0 | norec
def fwd():
    w["counters"][:2 * M + 5].zero_()          # the binning counters (a launch of its own: ~3 us, the same for every variant)
    hist = VARIANT != "nohist"
    _lib.call("so_preprocess_fwd", 1, N, eng.K, 0 if VARIANT == "deg0" else 3, p(s["means"].data), p(s["scales"].data), p(s["quats"].data), p(s["opacities"].data),
              p(s["sh0"].data), p(s["shN"].data), p(w["viewmats"]), p(w["Ks"]), W, H, 0.3, 0.01, 1e8, 0.0, 0, 0, 16,
              p(w["radii"]), p(w["means2d"]), p(w["depths"]), p(w["conics"]), p(w["opacities"]), p(w["colors"]),
              p(w["tiles_per_gauss"]), p(w["counters"]) if hist else 0, 0 if VARIANT == "norec" else p(w["rec"]), 0 if VARIANT == "norec" else p(w["vrec"]), 0, 0,
              0 if VARIANT == "nocull" else 1, p(w["key_buf"]) if hist else 0, eng.bin_capacity if hist else 0,
              p(w["counters"][2 * eng.M + 2:]) if hist else 0, _lib.stream())
def bwd():
    _lib.call("so_preprocess_bwd", 1, N, eng.K, 3, p(s["means"].data), p(s["scales"].data), p(s["quats"].data), p(s["opacities"].data),
              p(s["sh0"].data), p(s["shN"].data), p(w["viewmats"]), p(w["Ks"]), W, H, 0.3, 0, 0, p(w["radii"]), p(w["opacities"]),
              p(w["colors"]), 0, 0, 0, 0, 0, 0, 0.0, 0.0, p(g["means"]), p(g["scales"]), p(g["quats"]), p(g["opacities"]),
              p(g["sh0"]), p(g["shN"]), 0, 0, p(w["vrec"]), 0, 0, 0, 0, _lib.stream())
for name, fn in (("preprocess_fwd", fwd), ("preprocess_bwd", bwd)):
    for reps in (1, 20):
        ts = []
        for trial in range(5):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        print(f"{name}: {reps:2d} back-to-back launches -> {min(ts):.1f} us per launch (min of 5), median {sorted(ts)[2]:.1f}")
