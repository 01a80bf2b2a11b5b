"""BASELINE.json configs[0] ("c1"): 10k random Gaussians, one 256x256 pinhole camera, the CPU oracle only (no GPU): forward,
forward + backward and a full iteration (torch Adam), >= 5 warm-up + >= 20 timed, median -- BASELINE.md section 2.  Run on the
GPU box's host cores for the reported number (the thread count is printed); anywhere else it is a sanity number."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import c_oracle as CO, torch_oracle as O  # noqa: E402
from oracle.ssim_oracle import photometric_loss  # noqa: E402
from splat_one_amd.scene import make_scene  # noqa: E402

threads = min(os.cpu_count() or 1, 64)
torch.set_num_threads(threads)
os.environ["OMP_NUM_THREADS"] = str(threads)
W = H = 256
out = {"threads": threads, "config": "c1: 10k Gaussians, 256x256, SH degree 3, torch fp32 + oracle/c/raster_oracle.c (f32, OpenMP)"}
for regime in ("mcmc", "ref"):
    splats, c2w, Ks = make_scene(10_000, W, H, regime=regime)
    p = {k: v.clone().requires_grad_(True) for k, v in splats.items()}
    opt = torch.optim.Adam(p.values(), lr=1e-3, eps=1e-15)
    vm = torch.linalg.inv(c2w)
    px = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(7))
    raster = CO.raster_fn()

    def fwd():
        return O.rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), torch.cat([p["sh0"], p["shN"]], 1),
                               vm, Ks, W, H, sh_degree=3, near_plane=0.01, far_plane=1e8, raster_fn=raster, dtype=torch.float32)

    res = {}
    for what in ("fwd", "fwd_bwd", "iteration"):
        ts = []
        for i in range(25):
            t0 = time.time()
            if what == "fwd":
                with torch.no_grad():
                    rc, _, meta = fwd()
            else:
                rc, _, meta = fwd()
                loss, _, _ = photometric_loss(rc, px, 0.2, dtype=torch.float32)
                loss.backward()
                if what == "iteration":
                    opt.step()
                opt.zero_grad(set_to_none=True)
            if i >= 5:
                ts.append(time.time() - t0)
        med = sorted(ts)[len(ts) // 2]
        res[what] = {"ms": med * 1e3, "it_s": 1.0 / med}
    res["forward_mpix_per_s"] = W * H / (res["fwd"]["ms"] * 1e-3) / 1e6
    res["tile_intersections"] = int(meta["flatten_ids"].numel()) if "flatten_ids" in meta else None
    out[regime] = res
print(json.dumps(out))
