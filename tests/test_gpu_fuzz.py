"""Seeded random configurations of `rasterization` against the float64 oracle: camera model x tile size x ragged image
sizes x render mode x antialiasing x background x packed x number of cameras x SH degree, small scenes (the oracle
finishes each in well under a second).  What the fixed-size parity tests do not reach: combinations."""
import math
import os
import random

import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as O
from splat_one_amd.scene import lookat_c2w
from tests.util import small_scene

pytestmark = pytest.mark.gpu


def _case(seed):
    rnd = random.Random(seed)
    model = rnd.choice(["pinhole", "pinhole", "fisheye", "ortho", "spherical"])
    tile = rnd.choice([16, 16, 8])
    if model == "spherical":
        W, H = rnd.choice([(128, 64), (96, 48), (100, 50), (160, 80)])       # 100: not a multiple of the tile size (no wrap)
    else:
        W, H = rnd.randint(33, 150), rnd.randint(25, 110)
    C = rnd.choice([1, 1, 2, 3])
    N = rnd.randint(50, 700)
    deg = rnd.choice([0, 1, 2, 3, 3])
    mode = rnd.choice(["RGB", "RGB", "RGB+ED", "RGB+D", "D", "ED"])
    aa = rnd.random() < 0.3
    bg = rnd.random() < 0.4
    packed = rnd.random() < 0.35
    scale = rnd.choice([0.08, 0.2, 0.5])
    static = (not packed) and rnd.random() < 0.3          # the sync-free binning into caller-sized buffers
    return dict(model=model, tile=tile, W=W, H=H, C=C, N=N, deg=deg, mode=mode, aa=aa, bg=bg, packed=packed, scale=scale,
                static=static)


def _operator_against_the_oracle(dev, seed):
    """-> (cfg, (device image, oracle image), (device alpha, oracle alpha), device gradients, oracle gradients): the two legs of the
    operator fuzz test (also tools/dbg_fuzz_one.py --operator)."""
    from splat_one_amd import rasterization
    cfg = _case(seed)
    W, H, C, N = cfg["W"], cfg["H"], cfg["C"], cfg["N"]
    means, quats, scales, opac, sh = small_scene(N=N, seed=100 + seed, scale=cfg["scale"])
    g = torch.Generator().manual_seed(1000 + seed)
    if cfg["model"] == "spherical":      # cameras inside the cloud
        c2w = torch.eye(4)[None].repeat(C, 1, 1)
        c2w[:, :3, 3] = torch.randn(C, 3, generator=g) * 0.4
    else:
        c2w = torch.stack([lookat_c2w((6.0 * math.sin(0.9 * i + 0.3 * seed), 0.5 * i - 0.4, -6.0 * math.cos(0.9 * i + 0.3 * seed)))
                           for i in range(C)])
    viewmats = torch.linalg.inv(c2w).contiguous()
    f = float(max(W, H)) * (0.04 if cfg["model"] == "ortho" else 0.9)
    Ks = torch.tensor([[f, 0, W / 2.0], [0, f * 1.1, H / 2.0], [0, 0, 1]])[None].repeat(C, 1, 1)
    X = {"RGB": 3, "RGB+ED": 4, "RGB+D": 4, "D": 1, "ED": 1}[cfg["mode"]]
    bg = torch.rand(C, 3, generator=g) if (cfg["bg"] and cfg["mode"].startswith("RGB")) else None
    w_rgb, w_a = torch.rand(C, H, W, X, generator=g), torch.rand(C, H, W, 1, generator=g)
    kw = dict(sh_degree=cfg["deg"], tile_size=cfg["tile"], render_mode=cfg["mode"], camera_model=cfg["model"],
              rasterize_mode="antialiased" if cfg["aa"] else "classic", near_plane=0.01, far_plane=1e8)
    K = (cfg["deg"] + 1) ** 2

    def run(fn, to, dt, **extra):
        p = [t.detach().clone().to(to).to(dt).requires_grad_(True) for t in (means, quats, scales, opac, sh[:, :K].contiguous())]
        rc, ra, meta = fn(*p, viewmats.to(to).to(dt), Ks.to(to).to(dt), W, H,
                          backgrounds=None if bg is None else bg.to(to).to(dt), **kw, **extra)
        ((rc * w_rgb.to(rc)).sum() + (ra * w_a.to(ra)).sum()).backward()
        grads = [(torch.zeros_like(t) if t.grad is None else t.grad).detach().cpu().double() for t in p]   # (depth-only modes: no SH gradient)
        return rc.detach().cpu().double(), ra.detach().cpu().double(), grads, meta

    extra = {"isect_capacity": 400_000, "workspace": {}} if cfg["static"] else {}
    rc_h, ra_h, g_h, m_h = run(rasterization, dev, torch.float32, packed=cfg["packed"], **extra)
    if cfg["static"]:
        assert int(m_h["isect_overflow"].item()) == 0 and 0 < int(m_h["n_isects"].item()) <= 400_000
    # the oracle sorts by the device's float32 depth keys
    if cfg["packed"]:
        keys = torch.full((C, N), 1e30)
        keys[m_h["camera_ids"].cpu(), m_h["gaussian_ids"].cpu()] = m_h["depths"].detach().cpu()
    else:
        keys = m_h["depths"].detach().cpu().clone()
    rc_o, ra_o, g_o, m_o = run(O.rasterization, "cpu", torch.float64, raster_fn=CO.raster_fn(), sort_depths=keys)
    assert int((m_o["radii"] > 0).sum()) > 0, cfg
    return cfg, (rc_h, rc_o), (ra_h, ra_o), g_h, g_o


@pytest.mark.parametrize("seed", list(range(36)))
def test_random_configuration_against_the_oracle(dev, seed):
    cfg, (rc_h, rc_o), (ra_h, ra_o), g_h, g_o = _operator_against_the_oracle(dev, seed)
    # depth channels carry world units (up to ~10): the per-pixel bar scales with the channel's magnitude
    scale_px = max(1.0, float(rc_o.abs().max()))
    assert (rc_h - rc_o).abs().mean().item() <= 1e-4 * scale_px, (cfg, (rc_h - rc_o).abs().mean().item())
    assert (ra_h - ra_o).abs().mean().item() <= 1e-4, cfg
    names = ["means", "quats", "scales", "opacities", "sh"]
    ref_scale = g_o[2].norm().item()
    # ONE bar for every camera model (north_star: gradients <= 1e-3 relative).  Rounds 1-2 allowed fisheye / spherical
    # cases 3e-3; 120 seeds of this test and 120 of the engine test below were run at a uniform 1e-3 on the round-3 library
    # (tools/gpu_fuzz_bar.sh -> all pass), so the exception is gone.
    bar = float(os.environ.get("SPLAT_ONE_AMD_FUZZ_BAR", 1e-3))
    for k, a, b in zip(names, g_h, g_o):
        floor = 1e-5 * ref_scale if k == "quats" else 1e-9
        assert (a - b).norm().item() <= bar * b.norm().item() + floor, (cfg, k, (a - b).norm().item(), b.norm().item())


def test_image_shapes_of_equal_tile_count_do_not_share_bins(dev):
    """Seeds 2222 (one 145 x 33 view) and 2344 (two 71 x 41 views) are 30 tiles each.  While the bins of rasterization() were kept
    per tile COUNT the second inherited the first's measured lists (fullest 14 -> 1024 slots), skipped its own measuring call
    and cut its 1400-entry list: forward mean |diff| 8e-4 in the run 2201..2600 of the fuzzer, fine alone (tools/dbg_order_small.py).
    They are kept per grid (C, tile_w, tile_h) now."""
    from splat_one_amd import raster_op
    raster_op._BINS.clear()
    for seed in (2222, 2344):
        cfg, (rc_h, rc_o), (ra_h, ra_o), g_h, g_o = _operator_against_the_oracle(dev, seed)
        assert (rc_h - rc_o).abs().mean().item() <= 1e-4, (cfg, (rc_h - rc_o).abs().mean().item())
        assert (ra_h - ra_o).abs().mean().item() <= 1e-4, cfg
    grids = [k[3] for k in raster_op._BINS if k[2] == 30]
    assert sorted(grids) == [(1, 10, 3), (2, 5, 3)], grids
    assert raster_op.pending_overflow() == 0


def _engine_case(seed):
    rnd = random.Random(7000 + seed)
    C = rnd.choice([1, 1, 2, 3])
    models = [rnd.choice(["pinhole", "pinhole", "fisheye", "spherical"]) for _ in range(C)]
    if "spherical" in models:
        W, H = rnd.choice([(128, 64), (160, 80), (120, 60)])
    else:
        W, H = rnd.randint(40, 160), rnd.randint(30, 110)
    case = dict(C=C, models=models, W=W, H=H, N=rnd.randint(200, 3000), deg=rnd.choice([0, 1, 2, 3, 3]),
                aa=rnd.random() < 0.3, binned=rnd.random() < 0.6, tile_cull=rnd.random() < 0.7,
                f16=rnd.random() < 0.2, regs=rnd.random() < 0.4, scale=rnd.choice([0.1, 0.25, 0.6]),
                lam=rnd.choice([0.2, 0.2, 0.0, 0.5]))
    if "spherical" in models:
        # a panorama camera sits INSIDE the cloud: with image-filling splats hundreds of pixels lie within float32 rounding
        # of the alpha = 1/255 contour, each worth 4e-3 of a pixel whichever way it falls -- keep those splats moderate
        case["scale"] = min(case["scale"], 0.25)
    return case


def _engine_against_the_oracle(dev, seed):
    """-> (cfg, engine gradients, oracle gradients (None where the oracle has none), K, forward |difference| map, (engine losses,
    oracle L1, oracle SSIM loss)): the two legs of the engine fuzz test (also tools/dbg_fuzz_one.py)."""
    from oracle import ssim_oracle as SSO
    from splat_one_amd.engine import FusedEngine
    from splat_one_amd.trainer import Config, Runner
    cfg = _engine_case(seed)
    C, W, H, N = cfg["C"], cfg["W"], cfg["H"], cfg["N"]
    g = torch.Generator().manual_seed(9000 + seed)
    r = Runner(0, 0, 1, Config(init_num_pts=N, init_scale=cfg["scale"] * 3.0, init_opa=0.4, shN_init_std=0.15, batch_size=C),
               scene_scale=1.0 / 1.1)
    with torch.no_grad():
        r.splats["scales"].add_((torch.randn(N, 3, generator=g) * 0.4).to(dev))
        r.splats["quats"].copy_(torch.randn(N, 4, generator=g).to(dev))
    c2w = torch.stack([torch.eye(4) if m == "spherical" else lookat_c2w((7.0 * math.sin(1.1 * i + seed), 0.6 * i - 0.5, -7.0 * math.cos(1.1 * i + seed)))
                       for i, m in enumerate(cfg["models"])])
    for i, m in enumerate(cfg["models"]):
        if m == "spherical":
            c2w[i, :3, 3] = torch.randn(3, generator=g) * 0.5
    f = 0.9 * max(W, H)
    Ks = torch.tensor([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]])[None].repeat(C, 1, 1)
    pixels = torch.rand(C, H, W, 3, generator=g)
    oreg, sreg = (0.01, 0.02) if cfg["regs"] else (0.0, 0.0)
    eng = FusedEngine(r.splats, r.optimizers, W, H, C, sh_degree=cfg["deg"], camera_model=cfg["models"], use_graph=False,
                      antialiased=cfg["aa"], binned=cfg["binned"], tile_cull=cfg["tile_cull"], ssim_lambda=cfg["lam"],
                      opacity_reg=oreg, scale_reg=sreg, attr_dtype="f16" if cfg["f16"] else "f32")
    eng.set_views(c2w.to(dev), Ks.to(dev), pixels.to(dev))
    eng.fwd_bwd()
    assert eng.stats()["overflow"] == 0, cfg
    g_eng = {k: v.grad.detach().cpu().double() for k, v in r.splats.items()}
    loss_eng = eng.loss().cpu()
    # oracle, view by view (per-view camera models), at the half-rounded attributes when the engine reads float16 rows
    p = {k: v.detach().cpu().double() for k, v in r.splats.items()}
    if cfg["f16"]:
        for k in ("quats", "scales", "sh0", "shN"):
            p[k] = p[k].float().half().double()
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    K = (cfg["deg"] + 1) ** 2
    colors = torch.cat([p["sh0"], p["shN"]], 1)[:, :K]
    # the tile-sort keys are the float32 bit patterns of the depths: the oracle sorts by the DEVICE's (DESIGN.md section 3;
    # a panorama's depth is a range, one square root further from exact than a pinhole's z, and ties are everywhere
    # when the camera sits inside the cloud)
    dev_depths = eng.ws["depths"].detach().cpu().reshape(C, N).clone()
    rcs = []
    for c in range(C):
        rc, _, _ = O.rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), colors,
                                   torch.linalg.inv(c2w[c:c + 1]).double(), Ks[c:c + 1].double(), W, H, sh_degree=cfg["deg"],
                                   near_plane=0.01, far_plane=1e8, camera_model=cfg["models"][c],
                                   rasterize_mode="antialiased" if cfg["aa"] else "classic", raster_fn=CO.raster_fn(),
                                   sort_depths=dev_depths[c:c + 1])
        rcs.append(rc)
    rc = torch.cat(rcs)
    # the L1 term is differentiated with the DEVICE's sign pattern: where |render - target| is below the float32 error of the
    # render the two precisions legitimately take opposite signs -- in these small images one such
    # pixel is worth percent of a gradient norm when the loss is mostly L1
    signs = torch.sign(eng.ws["render_colors"].cpu().double() - pixels.double())
    loss_o, l1_o, ss_o = SSO.photometric_loss(rc, pixels.double(), cfg["lam"], l1_signs=signs)
    loss_o = loss_o + oreg * torch.sigmoid(p["opacities"]).abs().mean() + sreg * torch.exp(p["scales"]).abs().mean()
    loss_o.backward()
    fwd = (eng.ws["render_colors"].cpu().double() - rc.detach()).abs()
    return cfg, g_eng, {k: p[k].grad for k in p}, K, fwd, (loss_eng, l1_o, ss_o)


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_engine_configuration_against_the_oracle(dev, seed):
    """The fused engine (one training iteration's forward + loss + backward) on random combinations of per-view camera
    models, view counts, image sizes, SH degree, antialiasing, list layout, tile culling, float16 rows and regularisers."""
    cfg, g_eng, g_ref, K, fwd, (loss_eng, l1_o, ss_o) = _engine_against_the_oracle(dev, seed)
    assert fwd.mean().item() <= 1e-4, (cfg, "forward L1", fwd.mean().item(), "max", fwd.max().item(), "pixels > 1e-3", int((fwd > 1e-3).sum()))
    assert abs(loss_eng[1].item() - l1_o.item()) < 2e-5 and abs(loss_eng[2].item() - ss_o.item()) < 2e-5, \
        (cfg, "loss", loss_eng.tolist(), l1_o.item(), ss_o.item())
    bar = float(os.environ.get("SPLAT_ONE_AMD_FUZZ_BAR", 1e-3))     # every camera model: the north_star bar
    for k in g_eng:
        ref = g_ref[k]
        if ref is None:
            assert not g_eng[k].any(), (cfg, k)
            continue
        if k == "shN" and K < 16:
            assert not g_eng[k][:, K - 1:].any(), (cfg, k)
        floor = 1e-5 * g_ref["scales"].norm().item() if k == "quats" else 1e-12
        assert (g_eng[k] - ref).norm().item() <= bar * ref.norm().item() + floor, (cfg, k, (g_eng[k] - ref).norm().item(), ref.norm().item())
