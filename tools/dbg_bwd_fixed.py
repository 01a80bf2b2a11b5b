"""Per-stage times of the fused step on a FIXED model state (no optimiser step between launches): product library against
ablation / variant builds (SPLAT_ONE_AMD_LIB) see the same lists.  usage: dbg_bwd_fixed.py [N W H [regime [train_steps]]]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd import _lib
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.trainer import Config, Runner
from splat_one_amd.engine import FusedEngine
N, W, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (100_000, 1920, 1080)
regime = sys.argv[4] if len(sys.argv) > 4 else "mcmc"
dev = torch.device("cuda:0")
sc, op = (1.0, 0.1) if regime == "ref" else (0.1, 0.5)
r = Runner(0, 0, 1, Config(init_num_pts=N, init_scale=sc, init_opa=op, shN_init_std=0.1, sh_degree_interval=1), scene_scale=1 / 1.1)
ring = ring_cameras(8).to(dev)
Ks = pinhole_K(W, H)[None].to(dev)
tg = [torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(100 + v)).to(dev) for v in range(8)]
eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False)
for i in range(3):          # warm-up (module load, bin probe)
    eng.set_views(ring[i:i + 1], Ks, tg[i])
    if os.environ.get("SPLAT_ONE_AMD_BWD_TILE") in ("0", "1"):      # override the engine's choice of backward rasteriser
        eng.cfg["raster_impl"] = int(os.environ["SPLAT_ONE_AMD_BWD_TILE"])
    eng.fwd_bwd()
torch.cuda.synchronize()
_lib.call("so_profile_enable", 1)
for rep in range(5):
    for v in range(8):
        eng.set_views(ring[v:v + 1], Ks, tg[v]); eng.fwd_bwd()
prof = _lib.stage_profile()
_lib.call("so_profile_enable", 0)
st = eng.stats()
print(json.dumps({"lib": os.environ.get("SPLAT_ONE_AMD_LIB", "product"), "N": N, "WxH": [W, H], "regime": regime, "n_isects_last_view": st["n_isects"],
                  "us": {k: round(1e3 * v[1], 1) for k, v in prof.items() if v[0]}}))
