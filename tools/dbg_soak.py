"""Soak of both training paths with the list check on (SPLAT_ONE_AMD_CHECK_LISTS=1 makes the operator-level binning verify
every slot of its exact-size lists before the rasteriser runs): rotating views, densification every 100 iterations,
several camera models / layouts.  Prints one line per configuration; exits non-zero on the first failure."""
import os, sys, time, warnings
os.environ["SPLAT_ONE_AMD_CHECK_LISTS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.strategy import DefaultStrategy
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
W, H, N, STEPS = 640, 352, 60000, int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cams = ring_cameras(8).to(dev)
inside = torch.eye(4)[None].repeat(8, 1, 1)
inside[:, :3, 3] = torch.randn(8, 3, generator=torch.Generator().manual_seed(3)) * 0.3
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
targets = [torch.stack([(xx + 0.1 * v) % 1, (yy + 0.05 * v) % 1, 0.5 * (xx + yy)], -1)[None].to(dev).contiguous() for v in range(8)]
Ks = pinhole_K(W, H)[None].to(dev)
CASES = [("operator dense pinhole", dict(fused=False)),
         ("operator packed sparse_grad", dict(fused=False, packed=True, sparse_grad=True)),
         ("operator antialiased fisheye", dict(fused=False, camera_model="fisheye", antialiased=True)),
         ("operator spherical (periodic)", dict(fused=False, camera_model="spherical")),
         ("engine binned", dict(fused=True)),
         ("engine compact lists", dict(fused=True, binned=False)),
         ("engine spherical", dict(fused=True, camera_model="spherical")),
         ("engine f16 rows (device refinement)", dict(fused=True, attr_dtype="f16")),
         ("engine two-kernel Adam", dict(fused=True, fuse_adam=False)),
         ("engine no tile cull", dict(fused=True, tile_cull=False)),
         ("engine antialiased", dict(fused=True, antialiased=True)),
         ("engine host-side refinement", dict(fused=True, device_refine=False)),
         ("engine absgrad + revised opacity", dict(fused=True, _strategy=dict(absgrad=True, revised_opacity=True, grow_grad2d=0.0008))),
         ("engine batch of 2 (pinhole + fisheye)", dict(fused=True, batch_size=2, camera_model=["pinhole", "fisheye"])),
         ("engine MCMC", dict(fused=True, _mcmc=True, opacity_reg=0.01, scale_reg=0.01)),
         ("operator MCMC", dict(fused=False, _mcmc=True, opacity_reg=0.01, scale_reg=0.01)),
         ("operator visible_adam", dict(fused=False, visible_adam=True)),
         ("operator batch of 2", dict(fused=False, batch_size=2))]
bad = 0
for name, kw in CASES:
    kw = dict(kw)
    skw = kw.pop("_strategy", {})
    if kw.pop("_mcmc", False):
        from splat_one_amd.strategy import MCMCStrategy
        strat = MCMCStrategy(cap_max=120000, refine_start_iter=100, refine_every=100, verbose=False)
    else:
        strat = DefaultStrategy(refine_start_iter=100, refine_every=100, reset_every=600, verbose=False, **skw)
    cfg = Config(init_num_pts=N, init_scale=0.3, init_opa=0.3, sh_degree_interval=200, strategy=strat, shN_init_std=0.05, **kw)
    r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
    c2ws = inside.to(dev) if kw.get("camera_model") == "spherical" else cams
    t0 = time.time()
    try:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            B = int(kw.get("batch_size", 1))
            for step in range(STEPS):
                v = (step * 3) % 8
                if B == 1:
                    loss = r.train_step(c2ws[v:v + 1], Ks, targets[v])
                else:
                    vs = [(v + j) % 8 for j in range(B)]
                    loss = r.train_step(c2ws[vs].contiguous(), Ks.repeat(B, 1, 1), torch.cat([targets[j] for j in vs]))
            torch.cuda.synchronize()
        ok = all(torch.isfinite(p).all().item() for p in r.splats.values()) and bool(torch.isfinite(loss).all())
        print(f"{name:38s} N {N} -> {len(r.splats['means'])}  loss {float(loss):.4f}  finite {ok}  warnings {len(rec)}  {time.time() - t0:.1f} s", flush=True)
        bad += not ok
    except Exception as e:   # noqa: BLE001
        print(f"{name:38s} FAILED at step {r.step}: {str(e)[:600]}", flush=True)
        bad += 1
    del r
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
