"""Experiment (round 3): does launching the heaviest tiles first shorten the rasteriser kernels' tail?
Needs the variant library built with -DSO_TILE_PERM_EXPERIMENT (tools/build_lib_variant.sh perm "-DSO_TILE_PERM_EXPERIMENT"),
loaded through SPLAT_ONE_AMD_LIB.  c2 workload of bench.py (100k Gaussians, 1080p, 8 ring views); per view the tile order
is computed on the host from the list lengths of a warm-up step and copied into the table before each step."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from splat_one_amd import _lib                                   # noqa: E402
from splat_one_amd.scene import pinhole_K, ring_cameras          # noqa: E402
from splat_one_amd.trainer import Config, Runner                 # noqa: E402

dev = torch.device("cuda:0")
W, H, N, NV = 1920, 1080, 100_000, 8
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, batch_size=1, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
ring = ring_cameras(NV)
g = torch.Generator().manual_seed(100)
targets = [torch.rand(1, H, W, 3, generator=g).to(dev) for _ in range(NV)]
K1 = pinhole_K(W, H)[None].to(dev)
cams = [ring[v:v + 1].contiguous().to(dev) for v in range(NV)]


def run(n, hook=None):
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(n):
        if hook:
            hook(i % NV)
        r.train_step(cams[i % NV], K1, targets[i % NV])
    torch.cuda.synchronize()
    return n / (time.time() - t0)


run(24)
counts = []
for v in range(NV):
    r.train_step(cams[v], K1, targets[v])
    torch.cuda.synchronize()
    counts.append(r._engine.ws["counters"][:r._engine.M].clone())
M = counts[0].numel()
perms = [torch.argsort(c, descending=True, stable=True).to(torch.int32) for c in counts]
table = torch.arange(M, dtype=torch.int32, device=dev)
out = {"tiles": M, "max_list": [int(c.max()) for c in counts], "mean_list": [float(c.float().mean()) for c in counts]}
lib = _lib.load()
eng = r._engine
eng.use_graph = False


def kernels(hook=None, n=200):
    """stage timers (HIP events around every kernel of the un-captured step), mean us"""
    run(16, hook)
    lib.so_profile_enable(1)
    run(n, hook)
    prof = _lib.stage_profile()
    lib.so_profile_enable(0)
    return {k: round(v[1] * 1e3, 1) for k, v in prof.items() if "rasterize" in k}


out["xcd_runs_of_8"] = [kernels(), kernels()]
for name in ("fwd", "bwd", "both"):
    for k in ("fwd", "bwd"):
        fn = getattr(lib, "so_debug_tile_perm_" + k)
        fn.argtypes = [ctypes.c_void_p]
        assert fn(table.data_ptr() if name in (k, "both") else None) == 0
    out[f"heaviest_first_{name}"] = [kernels(lambda v: table.copy_(perms[v])) for _ in range(2)]
print(json.dumps(out))
