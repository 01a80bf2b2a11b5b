"""The fused engine (raw parameters -> gradients -> Adam in two C-ABI calls, replayed as a hipGraph)
against (1) the operator-level path it must equal and (2) the float64 oracle."""
import copy

import pytest
import torch

from oracle import c_oracle as CO
from oracle import ssim_oracle as SSO
from oracle import torch_oracle as O
from splat_one_amd.scene import make_scene
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _make(dev, N, W, H, regime, C=1, anisotropic=True, **cfgkw):
    from splat_one_amd.trainer import Config, Runner
    cfg = Config(init_num_pts=N, init_scale=(1.0 if regime == "ref" else 0.1), init_opa=(0.1 if regime == "ref" else 0.5),
                 shN_init_std=0.1, sh_degree_interval=1, **cfgkw)
    r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
    if anisotropic:
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            r.splats["scales"].add_((torch.randn(N, 3, generator=g) * 0.4).to(dev))
    from splat_one_amd.scene import front_camera, pinhole_K, ring_cameras
    c2w = (front_camera()[None] if C == 1 else ring_cameras(C)).to(dev)
    Ks = pinhole_K(W, H)[None].repeat(C, 1, 1).to(dev)
    pixels = torch.rand(C, H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    return r, c2w, Ks, pixels


@pytest.mark.parametrize("regime,C,kw", [("ref", 1, {}), ("mcmc", 2, {"opacity_reg": 0.01, "scale_reg": 0.01}),
                                         ("ref", 1, {"antialiased": True})])
def test_engine_gradients_match_oracle_and_operator_path(dev, regime, C, kw):
    from splat_one_amd.engine import FusedEngine
    N, W, H = 6000, 160, 96
    r, c2w, Ks, pixels = _make(dev, N, W, H, regime, C, **kw)
    r.step = 5                                   # SH degree 3
    st = r.cfg.strategy.initialize_state(1.0)
    eng = FusedEngine(r.splats, r.optimizers, W, H, C, sh_degree=3, strategy_state=st, use_graph=False,
                      antialiased=kw.get("antialiased", False), opacity_reg=kw.get("opacity_reg", 0.0),
                      scale_reg=kw.get("scale_reg", 0.0))
    eng.set_views(c2w, Ks, pixels)
    eng.fwd_bwd()
    g_eng = {k: v.grad.detach().clone().cpu().double() for k, v in r.splats.items()}
    loss_eng = eng.loss().cpu()
    stats = eng.stats()
    assert stats["overflow"] == 0 and stats["n_isects"] > 1000
    # float64 oracle on the same raw parameters
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in r.splats.items()}
    colors = torch.cat([p["sh0"], p["shN"]], 1)
    rc, ra, meta = O.rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), colors,
                                   torch.linalg.inv(c2w.cpu()), Ks.cpu(), W, H, sh_degree=3, near_plane=0.01, far_plane=1e8,
                                   rasterize_mode="antialiased" if kw.get("antialiased") else "classic",
                                   raster_fn=CO.raster_fn())
    meta["means2d"].retain_grad()
    loss_o, l1_o, ss_o = SSO.photometric_loss(rc, pixels.cpu(), 0.2)
    if kw.get("opacity_reg", 0) > 0:
        loss_o = loss_o + kw["opacity_reg"] * torch.sigmoid(p["opacities"].double()).abs().mean()
    if kw.get("scale_reg", 0) > 0:
        loss_o = loss_o + kw["scale_reg"] * torch.exp(p["scales"].double()).abs().mean()
    loss_o.backward()
    assert abs(loss_eng[1].item() - l1_o.item()) < 1e-5 and abs(loss_eng[2].item() - ss_o.item()) < 1e-5
    for k in g_eng:
        floor = 1e-5 * p["scales"].grad.norm().item() if k == "quats" else 0.0
        err = (g_eng[k] - p[k].grad.double()).norm().item()
        assert err <= 1e-3 * p[k].grad.norm().item() + floor, (k, err, p[k].grad.norm().item())
    # densification statistics == loop-free restatement from the oracle's means2d gradient
    g2d = meta["means2d"].grad
    sel = meta["radii"] > 0
    norms = torch.sqrt((g2d[..., 0] * W / 2 * C) ** 2 + (g2d[..., 1] * H / 2 * C) ** 2)
    assert rel_err(st["grad2d"], (norms * sel).sum(0)) < 2e-3
    assert torch.equal(st["count"].cpu(), sel.sum(0).float())
    # operator-level path of the product on the same parameters
    from splat_one_amd.losses import photometric_loss
    for v in r.splats.values():
        v.grad = None
    renders, alphas, info = r.rasterize_splats(c2w, Ks, W, H, sh_degree=3, near_plane=0.01, far_plane=1e8)
    loss_p, _, _ = photometric_loss(renders, pixels, 0.2)
    if kw.get("opacity_reg", 0) > 0:
        loss_p = loss_p + kw["opacity_reg"] * torch.sigmoid(r.splats["opacities"]).abs().mean()
    if kw.get("scale_reg", 0) > 0:
        loss_p = loss_p + kw["scale_reg"] * torch.exp(r.splats["scales"]).abs().mean()
    loss_p.backward()
    # the engine's lists are gsplat's minus the tiles its exact culling proves empty (test_engine_tile_cull_is_exact)
    assert stats["n_isects"] <= info["flatten_ids"].numel()
    # identical arithmetic; a 1-ulp opacity difference (fused sigmoid*comp) may flip one
    # alpha>=1/255 decision at a pixel, hence mean/max rather than bitwise
    assert (renders - eng.ws["render_colors"]).abs().mean().item() < 1e-6
    assert (renders - eng.ws["render_colors"]).abs().max().item() < 5e-3
    for k, v in r.splats.items():
        floor = 1e-5 * p["scales"].grad.norm().item() if k == "quats" else 0.0
        err = (v.grad.cpu().double() - g_eng[k]).norm().item()
        assert err <= 1e-3 * g_eng[k].norm().item() + floor, (k, err)


def test_engine_training_steps_match_runner_and_graph_replay(dev):
    """5 optimiser steps: hipGraph replay == eager engine == operator-level Runner.train_step."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 5000, 128, 96
    outs = []
    for mode in ("runner", "engine", "graph"):
        r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc", fused=(mode != "runner"))     # "runner": the operator-level step
        r.step = 10
        if mode == "runner":
            for _ in range(5):
                r.train_step(c2w, Ks, pixels)
        else:
            eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, strategy_state=r.strategy_state,
                              lr_gamma_means=r.lr_gamma, use_graph=(mode == "graph"))
            eng.steps_done = 0
            eng.set_views(c2w, Ks, pixels)
            for _ in range(5):
                eng.step()
            assert eng.stats()["overflow"] == 0
        outs.append({k: v.detach().clone() for k, v in r.splats.items()})
        if mode != "runner":
            assert float(r.optimizers["means"].state[r.splats["means"]]["step"]) == 5.0
    for k in outs[0]:
        # graph replay is the same launches as the eager engine: equal up to atomic ordering
        assert rel_err(outs[2][k], outs[1][k]) < 1e-5, k
        assert rel_err(outs[1][k], outs[0][k]) < 1e-3, k


def test_engine_render_forward_matches_operator_path(dev):
    from splat_one_amd.engine import FusedEngine
    N, W, H = 5000, 128, 96
    r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc", C=2)
    eng = FusedEngine(r.splats, r.optimizers, W, H, 2, sh_degree=3, use_graph=False)
    eng.set_cameras(c2w, Ks)
    rc, ra = eng.render()
    with torch.no_grad():
        rc2, ra2, _ = r.rasterize_splats(c2w, Ks, W, H, sh_degree=3, near_plane=0.01, far_plane=1e8)
    assert (rc - rc2).abs().max().item() < 1e-5 and (ra - ra2).abs().max().item() < 1e-5


@pytest.mark.parametrize("binned", [True, False])
def test_engine_survives_intersection_overflow(dev, binned):
    """Buffers sized far too small: the overflowing iterations stay in bounds, change nothing (optimiser and
    statistics skip on the device), are detected one step late without a device sync, the buffers grow, and the
    run continues exactly like one that had enough room from the start, minus the void iterations."""
    import warnings
    from splat_one_amd.engine import FusedEngine
    from splat_one_amd.scene import front_camera, pinhole_K
    from splat_one_amd.trainer import Config, Runner
    W, H, N, steps = 160, 120, 3000, 6

    def make(capacity):     # a total for the compact layout, turned into slots per tile for the binned one
        per_tile = None if capacity is None else max(16, capacity // 80)
        cfg = Config(init_num_pts=N, init_scale=0.6, init_opa=0.3, shN_init_std=0.05, sh_degree_interval=1, fused=True)
        r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
        with torch.no_grad():   # anisotropic: otherwise Adam amplifies a pure-rounding quaternion gradient
            r.splats["scales"].add_((torch.randn(N, 3, generator=torch.Generator().manual_seed(7)) * 0.4).to(dev))
        st = r.strategy_state
        eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, strategy_state=st, lr_gamma_means=r.lr_gamma,
                          isect_capacity=capacity, use_graph=True, binned=binned, bin_capacity=per_tile)
        return r, eng
    c2w = front_camera()[None].to(dev)
    Ks = pinhole_K(W, H)[None].to(dev)
    pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(5)).to(dev)

    r_small, e_small = make(512)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for _ in range(steps):
            e_small.set_views(c2w, Ks, pixels, schedule=True)
            e_small.step()
        e_small.set_views(c2w, Ks, pixels)          # one more staging: publishes the last iteration's status
        torch.cuda.synchronize()
    assert e_small.void_steps >= 1 and any("skipped" in str(w.message) for w in rec)
    assert e_small.capacity > 512 and e_small.stats()["overflow"] == 0
    done = steps - e_small.void_steps
    assert done >= 2 and e_small.steps_done == done
    assert float(r_small.optimizers["means"].state[r_small.splats["means"]]["step"]) == float(done)

    r_big, e_big = make(1 << 20)
    for _ in range(done):
        e_big.set_views(c2w, Ks, pixels, schedule=True)
        e_big.step()
    torch.cuda.synchronize()
    assert e_big.void_steps == 0
    for k in r_big.splats.keys():
        assert rel_err(r_small.splats[k], r_big.splats[k]) < 2e-4, k      # atomic-order noise through Adam
    assert rel_err(r_small.strategy_state["grad2d"], r_big.strategy_state["grad2d"]) < 1e-4
    assert torch.equal(r_small.strategy_state["count"], r_big.strategy_state["count"])

    # without an explicit capacity the first view is measured and the buffers sized from it: nothing is skipped
    r_auto, e_auto = make(None)
    if binned:                      # pretend the default guess was far too small
        e_auto.bin_capacity = 16
    else:
        e_auto.capacity = 1024
    for _ in range(2):
        e_auto.set_views(c2w, Ks, pixels, schedule=True)
        e_auto.step()
    torch.cuda.synchronize()
    assert e_auto.void_steps == 0 and (e_auto.bin_capacity > 16 if binned else e_auto.capacity > 1024)


def test_engine_long_binned_lists_equal_the_operator_lists(dev):
    """Tiles with thousands of Gaussians (gathered clouds, large scenes): the engine's binned lists -- keys in arrival order,
    sorted by the long-list kernel's depth buckets -- equal the operator path's compact lists of the same scene entry for
    entry (gsplat's order: depth, then Gaussian id), and the step trains without a void iteration."""
    from splat_one_amd.engine import FusedEngine
    from splat_one_amd.rendering import rasterization
    N, W, H = 14000, 48, 48
    r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False, tile_cull=False)
    eng.set_views(c2w, Ks, pixels)
    eng.fwd_bwd()
    st = eng.stats()
    offs, ids = eng.tile_lists()
    lens = [b - a for a, b in zip(offs[:-1], offs[1:])]
    assert st["overflow"] == 0 and max(lens) > 4096 and sum(1 for n in lens if n > 2048) >= 4, lens
    sp = r.splats
    with torch.no_grad():
        _, _, info = rasterization(sp["means"], sp["quats"], torch.exp(sp["scales"]), torch.sigmoid(sp["opacities"]),
                                   torch.cat([sp["sh0"], sp["shN"]], 1), torch.linalg.inv(c2w), Ks, W, H, sh_degree=3,
                                   near_plane=0.01, far_plane=1e8, packed=False, tile_cull=False)
    assert torch.equal(info["flatten_ids"].cpu(), ids.cpu())
    assert info["isect_offsets"].reshape(-1).cpu().tolist() == offs[:-1]


@pytest.mark.parametrize("regime,C,aa,binned", [("ref", 1, False, True), ("mcmc", 2, True, True), ("ref", 1, False, False)])
def test_engine_tile_cull_is_exact(dev, monkeypatch, regime, C, aa, binned):
    """Exact tile culling drops (Gaussian, tile) pairs that cannot reach alpha = 1/255 at any pixel of the tile:
    fewer intersections, the SAME image bit for bit, the same gradients up to the order of the atomic sums; the
    surviving lists are subsequences of gsplat's lists (which the engine reproduces exactly with culling off)."""
    # (the same backward walk in both runs: list segments -- chosen from the list lengths, which the cull changes -- take another
    # float32 route to the same gradients, 1e-5 ... 3e-5 apart; test_backward_in_list_segments_gives_the_same_gradients)
    monkeypatch.setenv("SPLAT_ONE_AMD_BWD_SEGMENTS", "1")
    from splat_one_amd.engine import FusedEngine
    from splat_one_amd.rendering import rasterization
    N, W, H = 8000, 200, 136
    res = {}
    for cull in (False, True):
        r, c2w, Ks, pixels = _make(dev, N, W, H, regime, C, antialiased=aa)
        with torch.no_grad():                    # some Gaussians that can never reach 1/255, some barely
            r.splats["opacities"][:500] = -6.0
            r.splats["opacities"][500:1000] = -5.0
        eng = FusedEngine(r.splats, r.optimizers, W, H, C, sh_degree=3, use_graph=False, tile_cull=cull, antialiased=aa,
                          binned=binned)
        eng.set_views(c2w, Ks, pixels)
        eng.fwd_bwd()
        st = eng.stats()
        assert st["overflow"] == 0
        M = eng.M
        offs, ids = eng.tile_lists()
        ids = ids.cpu()
        assert offs[-1] == st["n_isects"] == ids.numel()
        res[cull] = dict(img=eng.ws["render_colors"].clone(), alpha=eng.ws["render_alphas"].clone(), n=st["n_isects"],
                         grads={k: v.grad.detach().clone() for k, v in r.splats.items()}, offs=offs, ids=ids,
                         loss=eng.loss().clone(), radii=eng.ws["radii"].clone())
        # the operator-level lists with the same switch are the engine's, bit for bit (off = gsplat semantics)
        sp = r.splats
        with torch.no_grad():
            _, _, info = rasterization(sp["means"], sp["quats"], torch.exp(sp["scales"]), torch.sigmoid(sp["opacities"]),
                                       torch.cat([sp["sh0"], sp["shN"]], 1), torch.linalg.inv(c2w), Ks, W, H, sh_degree=3,
                                       near_plane=0.01, far_plane=1e8, rasterize_mode="antialiased" if aa else "classic",
                                       packed=False, tile_cull=cull)
        # meta shows gsplat's lists whatever the switch; the kernels' own lists have the engine's length
        assert int(info["n_isects_kernel"]) == st["n_isects"]
        if not cull:
            assert torch.equal(info["flatten_ids"].cpu(), ids)
    a, b = res[False], res[True]
    assert b["n"] < 0.9 * a["n"], (a["n"], b["n"])
    assert torch.equal(a["img"], b["img"]) and torch.equal(a["alpha"], b["alpha"])
    assert (a["loss"] - b["loss"]).abs().max().item() < 1e-6         # same image; the loss sums are atomic (order varies)
    assert torch.equal(a["radii"], b["radii"])                       # visibility (densification) is not touched
    for k in a["grads"]:
        d = (a["grads"][k] - b["grads"][k]).norm().item()
        assert d <= 1e-5 * a["grads"][k].norm().item() + 1e-12, (k, d)
    for t in range(len(a["offs"]) - 1):                              # per tile: an order-preserving subsequence
        full = a["ids"][a["offs"][t]:a["offs"][t + 1]].tolist()
        kept = b["ids"][b["offs"][t]:b["offs"][t + 1]].tolist()
        it = iter(full)
        assert all(any(g == f for f in it) for g in kept), t


def test_engine_render_after_rebuild_keeps_cameras_and_counts_no_void_step(dev):
    """A rebuild (densification) reallocates the workspace: the staged cameras survive it, so a forward-only render
    right after shows the same view; and a render that overflows its bins enlarges them without taking a training
    iteration back."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 4000, 160, 96
    r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, lr_gamma_means=r.lr_gamma)
    eng.set_views(c2w, Ks, pixels, schedule=True)
    eng.step()
    img0 = eng.render()[0].clone()
    eng.rebuild()
    img1 = eng.render()[0].clone()
    assert torch.equal(img0, img1) and img0.abs().sum().item() > 0
    # a render on bins that are far too small, then training goes on: buffers grow, nothing is rolled back
    eng.bin_capacity = 16
    eng.render()
    assert eng.stats()["overflow"] == 1
    for _ in range(3):
        eng.set_views(c2w, Ks, pixels, schedule=True)
        eng.step()
    torch.cuda.synchronize()
    assert eng.void_steps == 0 and eng.steps_done == 4 and eng.bin_capacity > 16 and eng.stats()["overflow"] == 0
    assert float(r.optimizers["means"].state[r.splats["means"]]["step"]) == 4.0


@pytest.mark.parametrize("use_graph", [False, True])
def test_engine_fused_adam_equals_the_two_kernel_step(dev, use_graph):
    """step() with the schedule staged applies Adam inside the backward kernel (so_step_desc.fuse_adam): parameters and
    moments after 6 iterations -- through the SH ramp, with regularisers and densification statistics -- equal those of
    the backward + so_adam_step_dev pair (same arithmetic; only the order of the rasteriser's atomic sums differs)."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 5003, 128, 96                      # odd N: a partial last wave in the staged sweep
    out = {}
    for fuse in (False, True):
        r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc", 2, opacity_reg=0.01, scale_reg=0.01)
        st = r.cfg.strategy.initialize_state(1.0)
        eng = FusedEngine(r.splats, r.optimizers, W, H, 2, sh_degree=0, strategy_state=st, lr_gamma_means=r.lr_gamma,
                          use_graph=use_graph, fuse_adam=fuse, opacity_reg=0.01, scale_reg=0.01)
        for it in range(6):
            eng.set_sh_degree(min(it, 3))
            eng.set_views(c2w, Ks, pixels, schedule=True)
            eng.step()
        torch.cuda.synchronize()
        assert eng.stats()["overflow"] == 0 and eng.steps_done == 6
        out[fuse] = ({k: v.detach().clone() for k, v in r.splats.items()},
                     {k: (r.optimizers[k].state[r.splats[k]]["exp_avg"].clone(), r.optimizers[k].state[r.splats[k]]["exp_avg_sq"].clone())
                      for k in r.splats.keys()}, st["grad2d"].clone(), st["count"].clone(), eng.loss().clone())
    (pa, ma, g2a, cna, la), (pb, mb, g2b, cnb, lb) = out[False], out[True]
    for k in pa:
        assert rel_err(pa[k], pb[k]) < 2e-5, k
        assert rel_err(ma[k][0], mb[k][0]) < 1e-4 and rel_err(ma[k][1], mb[k][1]) < 1e-4, k
        assert not torch.equal(pb[k], _make(dev, N, W, H, "mcmc", 2)[0].splats[k].detach()), k     # it did train
    assert rel_err(g2a, g2b) < 1e-4 and torch.equal(cna, cnb) and (la - lb).abs().max().item() < 1e-5



@pytest.mark.parametrize("use_graph,attr_dtype,device_refine", [(False, "f32", False), (True, "f32", True), (False, "f16", True)])
def test_head_plus_row_chunks_equal_the_whole_step(dev, use_graph, attr_dtype, device_refine):
    """Data-parallel replicas run the iteration as so_train_step_head + so_train_step_bwd_rows per row chunk (so that a
    chunk's reduce-scatter runs under the next chunk's kernel).  On ONE set of gradient records the chunked backward equals
    the backward over all rows BIT FOR BIT (gradients, regulariser terms, densification statistics), whatever the cut
    (uneven chunks, a chunk past the live rows); against the one-call step of a second engine it agrees up to the order of
    the rasteriser's atomic adds."""
    from splat_one_amd.engine import FusedEngine
    N, W, H, C = 5000, 160, 96, 2
    res = {}
    for split in (False, True):
        r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc", C, opacity_reg=0.01, scale_reg=0.01)
        st = r.cfg.strategy.initialize_state(1.0)
        eng = FusedEngine(r.splats, r.optimizers, W, H, C, sh_degree=3, strategy_state=st, use_graph=use_graph, attr_dtype=attr_dtype,
                          opacity_reg=0.01, scale_reg=0.01, device_refine=device_refine, capacity=(8192 if device_refine else None))
        rows = 8192 if device_refine else N
        stats = eng.dstats if device_refine else st
        snap = lambda: ({k: v[:N].clone() for k, v in eng.ws["grads"].items()}, stats["grad2d"][:N].clone(), stats["count"][:N].clone())
        if not split:
            eng.set_views(c2w, Ks, pixels)
            eng.fwd_bwd()
            torch.cuda.synchronize()
            res[split] = snap() + (eng.loss().clone(),)
            continue
        for it in range(2):          # twice: the second pass replays the captured head graph
            eng.set_views(c2w, Ks, pixels)
            eng.fwd_bwd_head()
        for k in ("grad2d", "count"):
            stats[k].zero_()
        eng.bwd_rows(0, rows)                                     # all rows in one launch ...
        torch.cuda.synchronize()
        whole = snap()
        for v in eng.ws["grads"].values():
            v.fill_(float("nan"))
        for k in ("grad2d", "count"):
            stats[k].zero_()
        for a, b in ((0, 1024), (1024, 1088), (1088, 4992), (4992, rows)):     # ... and cut into chunks, on the same records
            eng.bwd_rows(a, b)
        torch.cuda.synchronize()
        cut = snap()
        for k in whole[0]:
            assert torch.equal(whole[0][k], cut[0][k]) and whole[0][k].abs().sum() > 0, k
        assert torch.equal(whole[1], cut[1]) and torch.equal(whole[2], cut[2]) and float(cut[2].sum()) > 0
        res[split] = cut + (eng.loss().clone(),)
    for k in res[False][0]:
        assert rel_err(res[True][0][k], res[False][0][k]) < 1e-5, k
    assert rel_err(res[True][1], res[False][1]) < 1e-5 and torch.equal(res[True][2], res[False][2])
    assert torch.allclose(res[True][3], res[False][3], rtol=1e-6, atol=0)


@pytest.mark.parametrize("regime,C,model", [("ref", 1, "pinhole"), ("mcmc", 2, "pinhole"), ("ref", 1, "spherical")])
def test_tile_wave_backward_equals_the_quadrant_wave_backward(dev, regime, C, model):
    """so_step_desc.raster_impl = 1: the backward rasteriser as ONE wave per 16x16 tile (csrc/rasterize_bwd_tile.hip: four
    pixels per lane, one reduction and one atomic per (tile, Gaussian)) -- the engine's choice for long lists.  Same
    per-pixel arithmetic as the wave-per-quadrant kernel, sums in another order: gradients agree to rounding, and with the
    float64 oracle at the usual bar.  The spherical case crosses the +-pi seam (periodic image)."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 6000, 256 if model == "spherical" else 176, 128 if model == "spherical" else 112
    grads = {}
    for impl in (0, 1):
        r, c2w, Ks, pixels = _make(dev, N, W, H, regime, C)
        if model == "spherical":
            c2w = torch.eye(4, device=dev)[None]
        eng = FusedEngine(r.splats, r.optimizers, W, H, C, sh_degree=3, use_graph=False, camera_model=model)
        eng.set_views(c2w, Ks, pixels)          # (the capacity probe picks the mapping by list length: override it after)
        eng.cfg["raster_impl"] = impl
        eng.fwd_bwd()
        torch.cuda.synchronize()
        assert eng.stats()["overflow"] == 0
        grads[impl] = {k: v.grad.detach().clone() for k, v in r.splats.items()}
    for k in grads[0]:
        assert rel_err(grads[1][k], grads[0][k]) < 2e-5, k
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in r.splats.items()}
    rc, _ra, _meta = O.rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                     torch.cat([p["sh0"], p["shN"]], 1), torch.linalg.inv(c2w.cpu()), Ks.cpu(), W, H, sh_degree=3,
                                     near_plane=0.01, far_plane=1e8, camera_model=model, raster_fn=CO.raster_fn())
    loss_o, _, _ = SSO.photometric_loss(rc, pixels.cpu(), 0.2)
    loss_o.backward()
    for k, v in p.items():
        floor = 1e-5 * p["scales"].grad.norm() if k == "quats" else 0.0
        assert ((grads[1][k].cpu().double() - v.grad).norm() / (v.grad.norm() + floor)).item() <= 1e-3, k


def test_bins_and_backward_rasteriser_follow_growing_lists_without_a_read_back(dev, monkeypatch):
    """A device-side refinement can multiply the per-tile lists (BASELINE configs[3]: 1M -> 1.8M Gaussians over the timed
    region, 145 -> 337 entries per tile).  so_step_inputs gathers the maximum and the sum of the list lengths while it zeroes
    the counters and publishes them to host-mapped memory one call later; from those the engine rebuilds its bins at 8x the
    fullest tile as soon as they hold less than 2x of it -- BEFORE a tile overflows: no void iteration -- and moves the
    backward to one wave per tile once the mean list reaches 256 entries.  No synchronising call in the steps that only look
    (torch's sync debug mode); `reprobe_capacity()` is the explicit, synchronising form."""
    from splat_one_amd import list_policy
    from splat_one_amd.engine import FusedEngine
    # (one wave per tile is for images of >= 3072 tiles; this test follows the policy on a small image)
    monkeypatch.setattr(list_policy, "MIN_TILES_FOR_TILE_WAVES", 0)
    monkeypatch.setattr(list_policy, "TILES_PER_UNEVENNESS", (1e-3, 1e-3))      # (... and lets lists be as uneven as on a large one)
    monkeypatch.setattr(list_policy, "TILE_WAVES_MIN_ENTRIES", (256.0 * 77, 192.0 * 77))   # (... and as if a mean of 256 entries were much work)
    N, W, H = 6000, 176, 112
    r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc")
    st = r.cfg.strategy.initialize_state(1.0)
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=True, device_refine=True, capacity=16384,
                      strategy_state=st, bin_capacity=512)

    def steps(n, no_sync=False):
        if no_sync:
            torch.cuda.set_sync_debug_mode("error")
        try:
            for _ in range(n):
                eng.set_views(c2w, Ks, pixels, schedule=True)
                eng.step()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()

    steps(4)
    ws0, fullest0 = eng.ws, eng._fullest_tile()
    assert eng.cfg["raster_impl"] == 0 and eng.void_steps == 0 and eng.bin_capacity == 512 and 2 * fullest0 < 512
    steps(4, no_sync=True)                                   # lists with >= 2x headroom: looked at, left alone, no sync
    # (what the host sees is two iterations old and the model trains: close to the current lists, not equal)
    assert eng.ws is ws0 and abs(int(eng._status[3]) - eng._fullest_tile()) <= 8
    assert abs(int(eng._status[4]) - eng.stats()["n_isects"]) <= 0.02 * eng.stats()["n_isects"]
    # the splats swell a little (what a split / duplicate round does to the lists, as one edit of the active set): the
    # fullest tile passes HALF the bin -- not the bin
    with torch.no_grad():
        eng.sets[eng.active]["p"]["scales"][:N] += 0.9
    steps(5)
    fullest1 = eng._fullest_tile()
    assert 256 < fullest1 <= 512, fullest1
    assert eng.void_steps == 0 and eng.stats()["overflow"] == 0 and eng.bin_capacity >= 8 * fullest1 > 512 and eng.ws is not ws0
    # and further: still inside the new bins; the mean list passes 256 entries -> one wave per tile in the backward
    with torch.no_grad():
        eng.sets[eng.active]["p"]["scales"][:N] += 1.6
    steps(5)
    assert eng.void_steps == 0 and eng.stats()["overflow"] == 0
    assert eng.stats()["n_isects"] / eng.M >= 256 and eng.cfg["raster_impl"] == 1
    # the explicit form: one forward pass and one read, at once
    eng.cfg["raster_impl"] = 0
    eng.reprobe_capacity()
    steps(1)
    assert eng.cfg["raster_impl"] == 1


def test_skewed_cloud_falls_back_to_compact_lists_instead_of_raising(dev):
    """A cloud gathered in a few tiles (what real captures look like next to the uniform benchmark cube): bins sized for the
    fullest tile would exceed the memory budget.  The engine switches to the compact slotted lists at its capacity probe --
    one warning, no void iteration, no RuntimeError -- and trains like an engine built with
    binned=False."""
    import warnings
    from splat_one_amd.engine import FusedEngine
    N, W, H = 20_000, 320, 192
    res = {}
    for mode in ("budget", "compact"):
        r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc")
        with torch.no_grad():
            r.splats["means"].mul_(0.08)                      # everything lands in the middle of the image
        M = (W // 16) * (H // 16)
        kw = dict(bin_budget_bytes=12 * M * 96) if mode == "budget" else dict(binned=False)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=True, **kw)
            for _ in range(4):
                eng.set_views(c2w, Ks, pixels, schedule=True)
                eng.step()
            torch.cuda.synchronize()
        st = eng.stats()
        assert st["overflow"] == 0 and eng.void_steps == 0 and eng.steps_done == 4 and not eng.binned
        if mode == "budget":
            assert eng.fell_back_to_compact and sum("falling back to the compact" in str(w.message) for w in caught) == 1
            fullest = int(torch.bincount(eng.tile_lists()[1].new_tensor(
                [i for i, (a, b) in enumerate(zip(eng.tile_lists()[0][:-1], eng.tile_lists()[0][1:])) for _ in range(b - a)])).max())
            assert 2 * fullest > 96                            # bins with headroom really did not fit the budget
        res[mode] = {k: v.detach().clone() for k, v in r.splats.items()}
    for k in res["budget"]:
        assert rel_err(res["budget"][k], res["compact"][k]) < 1e-4, k


@pytest.mark.parametrize("policy", ["grow", "raise", "defer"])
def test_overflow_policy_is_honoured_when_the_bins_are_at_their_budget(dev, policy):
    """with the bins already at `bin_budget_bytes` an overflow used to fall back to the compact lists BEFORE
    looking at `on_overflow` -- "raise" did not raise (and the host's step count drifted from the device's), "defer" forgot
    what the rank had seen.  Bins fixed at 16 slots = the budget, a view whose fullest tile needs more:
    grow: the void iterations are taken back, compact lists from then on, the run continues;  raise: RuntimeError;
    defer: nothing changes until the caller -- Runner._dp_check_void -- calls take_back, which then falls back."""
    import warnings
    from splat_one_amd.engine import FusedEngine
    N, W, H = 6000, 160, 96
    r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc")
    with torch.no_grad():
        r.splats["means"].mul_(0.3)
    M = (W // 16) * (H // 16)
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=True, bin_capacity=16, bin_budget_bytes=12 * M * 16)
    assert eng.bin_capacity == 16 and eng._bin_limit == 16
    eng.on_overflow = policy

    def steps(n):
        for _ in range(n):
            eng.set_views(c2w, Ks, pixels, schedule=True)
            eng.step()

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        if policy == "raise":
            with pytest.raises(RuntimeError, match="overflowed"):
                steps(4)
            return
        steps(4)
        torch.cuda.synchronize()
        if policy == "defer":
            assert eng.binned and eng.void_steps == 0 and eng._compact_pending and eng._local_overflow_seen > 16
            assert eng.steps_done == 4                       # nothing taken back yet: that is the caller's (collective) decision
            seen, needed = eng.local_overflow_recent()
            assert seen and needed > 16
            eng.take_back(4, needed, grow=seen)              # (every iteration so far was void on the device)
        assert not eng.binned and eng.fell_back_to_compact
        assert sum("falling back to the compact" in str(w.message) for w in caught) == 1
        void = eng.void_steps
        assert void >= 2 and eng.steps_done == 4 - void
        assert float(r.optimizers["means"].state[r.splats["means"]]["step"]) == float(eng.steps_done)
        assert int(eng._step_dev[0].item()) == eng.steps_done
        steps(3)
        eng.set_views(c2w, Ks, pixels)
        torch.cuda.synchronize()
    assert eng.void_steps == void and eng.steps_done == 4 - void + 3 and eng.stats()["overflow"] == 0
    assert int(eng._step_dev[0].item()) == eng.steps_done
    assert all(torch.isfinite(v).all() for v in r.splats.values())


def test_sort_in_the_rasteriser_prologue_gives_the_same_lists_images_and_gradients(dev):
    """so_step_desc.sort_in_rasteriser (round 5): the forward rasteriser's workgroups sort their own tile lists -- one wave in
    registers up to 256 keys, the workgroup up to 2048, a scratch-free rank sort beyond -- instead of the sort kernels.  Same
    lists bit for bit, same image bit for bit, gradients equal up to atomic order; all three in-kernel paths are walked (a
    cloud gathered in the middle of a small image makes tiles of 30 ... 3000+ keys)."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 30_000, 256, 160
    out = {}
    for fold in (False, True):
        r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
        with torch.no_grad():
            r.splats["means"].mul_(0.25)
        eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False, fuse_adam=False)
        eng.set_views(c2w, Ks, pixels)           # (capacity probe, kernel choice)
        eng.sort_fold_ok, eng._fold = fold, fold
        eng.fwd_bwd()
        torch.cuda.synchronize()
        assert eng._desc().sort_in_rasteriser == int(fold) and eng.binned
        offs, ids = eng.tile_lists()
        lens = torch.tensor(offs[1:]) - torch.tensor(offs[:-1])
        out[fold] = dict(offs=offs, ids=ids.clone(), img=eng.ws["render_colors"].clone(), last=eng.ws["last_ids"].clone(),
                         grads={k: v.grad.detach().clone() for k, v in r.splats.items()}, lens=lens)
    a, b = out[False], out[True]
    assert int(b["lens"].max()) > 2048 and int(((b["lens"] > 256) & (b["lens"] <= 2048)).sum()) > 0 and int(((b["lens"] > 0) & (b["lens"] <= 256)).sum()) > 0
    assert a["offs"] == b["offs"] and torch.equal(a["ids"], b["ids"])
    assert torch.equal(a["img"], b["img"]) and torch.equal(a["last"], b["last"])
    for k in a["grads"]:
        assert rel_err(b["grads"][k], a["grads"][k]) < 1e-5, k


def test_tile_order_is_a_permutation_longest_list_first_and_changes_nothing(dev):
    """so_step_desc.tile_order (round 5): k_tile_order's workgroup -> tile table is a permutation of the tiles in which list
    lengths never increase by more than one length class (a counting sort over 256 classes of fullest / 256 entries each; empty
    tiles in the last class), and a step that takes its tiles in that order gives the same image bit for bit and the same gradients up to
    atomic order -- with one wave per tile and with four."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 20_000, 320, 192
    for impl in (0, 1):
        res = {}
        for lpt in (False, True):
            r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
            with torch.no_grad():
                r.splats["means"].mul_(0.3)              # skewed: full tiles in the middle, empty ones at the rim
            eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False, fuse_adam=False)
            eng.set_views(c2w, Ks, pixels)
            eng.cfg["raster_impl"], eng._lpt, eng.tile_order_lpt = impl, lpt, True
            eng.fwd_bwd()
            torch.cuda.synchronize()
            offs, _ = eng.tile_lists()
            lens = torch.tensor(offs[1:]) - torch.tensor(offs[:-1])
            res[lpt] = dict(img=eng.ws["render_colors"].clone(), grads={k: v.grad.detach().clone() for k, v in r.splats.items()})
            if lpt:
                order = eng.ws["tile_order"].cpu().long()
                assert torch.equal(torch.sort(order).values, torch.arange(eng.M))                  # a permutation
                got = lens[order]
                cls = int(lens.max()) // 256 + 1                                                   # entries per length class
                assert bool((got[1:] <= got[:-1] + cls).all()), "lengths rise by more than one class along the order"
                n_empty = int((lens == 0).sum())
                assert int(got[0]) >= int(lens.max()) - cls and n_empty > 0 and bool((got[-n_empty:] < cls).all())   # (the last class: 0 ... cls - 1 entries)
        assert torch.equal(res[False]["img"], res[True]["img"])
        for k in res[False]["grads"]:
            assert rel_err(res[True]["grads"][k], res[False]["grads"][k]) < 1e-5, (impl, k)


@pytest.mark.parametrize("f16", [False, True])
def test_replicated_bin_counters_give_the_same_lists_images_and_gradients(dev, monkeypatch, f16):
    """so_step_desc.bin_replicas (round 5): on images of few tiles the projection kernel's workgroups bump one of R copies of a
    tile's counter and fill one of R slices of its bin; k_bins_gather closes the slices up before the sort.  R = 1 / 4 / 8 give the
    same lists bit for bit, the same image bit for bit, gradients equal up to atomic order -- on a cloud gathered in the middle
    of a small image (tiles of 0 ... 3000+ keys: empty slices, full slices, slices of one key)."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 30_000, 256, 160
    out = {}
    for R in (1, 4, 8):
        monkeypatch.setenv("SPLAT_ONE_AMD_BIN_REPLICAS", str(R))
        r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
        with torch.no_grad():
            r.splats["means"].mul_(0.25)
        # (float16 rows: with the device-resident count, which the replicated counters need where the per-view arrays exist)
        eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False, fuse_adam=False, attr_dtype="f16" if f16 else "f32",
                          device_refine=f16, capacity=(32768 if f16 else None))
        eng.set_views(c2w, Ks, pixels)
        eng.fwd_bwd()
        torch.cuda.synchronize()
        assert eng.binned and eng.bin_replicas == R and eng._desc().bin_replicas == (R if R > 1 else 0) and eng.stats()["overflow"] == 0
        if R > 1:
            assert not eng.ws["bin_sub_counts"].any()              # the step leaves the counter copies zeroed for the next one
        offs, ids = eng.tile_lists()
        # (last_ids are positions in flatten_ids = tile * bin_capacity + position in the tile's list, and the capacity is rounded to R)
        out[R] = dict(offs=offs, ids=ids.clone(), img=eng.ws["render_colors"].clone(), last=eng.ws["last_ids"].clone() % eng.bin_capacity,
                      grads={k: v[:N].clone() for k, v in eng.ws["grads"].items()}, n=eng.stats()["n_isects"])
        eng.fwd_bwd()                                              # a second iteration on the same counters
        torch.cuda.synchronize()
        assert torch.equal(eng.ws["render_colors"], out[R]["img"]) and eng.stats()["n_isects"] == out[R]["n"]
    lens = torch.tensor(out[1]["offs"][1:]) - torch.tensor(out[1]["offs"][:-1])
    assert int(lens.max()) > 2048 and int((lens == 0).sum()) > 0 and int(((lens > 0) & (lens < 8)).sum()) > 0
    for R in (4, 8):
        a, b = out[1], out[R]
        assert a["n"] == b["n"] and a["offs"] == b["offs"] and torch.equal(a["ids"], b["ids"])
        assert torch.equal(a["img"], b["img"]) and torch.equal(a["last"], b["last"])
        for k in a["grads"]:
            assert rel_err(b["grads"][k], a["grads"][k]) < 1e-5, (R, k)


def test_a_bin_slice_that_overflows_voids_the_iteration_and_the_bins_grow(dev, monkeypatch):
    """Replicated counters: a slice holds bin_capacity / R keys, and one that overflows (while the tile's total is well below the
    bin's capacity) must void the iteration and size the new bins by R x the fullest slice."""
    from splat_one_amd.engine import FusedEngine
    monkeypatch.setenv("SPLAT_ONE_AMD_BIN_REPLICAS", "8")
    N, W, H = 20_000, 256, 160
    r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False)
    eng.set_views(c2w, Ks, pixels)
    eng.step()
    torch.cuda.synchronize()
    fullest = eng._fullest_tile()
    # bins that hold the fullest tile 1.05 times over: its eight slices hold 0.13 of it each and the fuller ones overflow
    eng._bin_hint = None
    eng.bin_capacity = 0
    eng._bin_hint = max(64, int(1.05 * fullest) // 8 * 8)
    eng._build_workspace()
    eng._probe_capacity = False
    cap0 = eng.bin_capacity
    assert cap0 >= fullest and cap0 < 1.2 * fullest and eng.bin_replicas == 8
    eng.set_views(c2w, Ks, pixels)
    steps0, void0 = eng.steps_done, eng.void_steps
    for _ in range(4):
        eng.set_views(c2w, Ks, pixels)
        eng.step()
    torch.cuda.synchronize()
    eng.set_views(c2w, Ks, pixels)          # (the check of the last iterations)
    assert eng.void_steps > void0 and eng.bin_capacity > cap0, (eng.void_steps, eng.bin_capacity, cap0, fullest)
    assert eng.steps_done == steps0 + 4 - (eng.void_steps - void0)
    eng.step()
    torch.cuda.synchronize()
    assert eng.stats()["overflow"] == 0


@pytest.mark.parametrize("segments", [2, 4, 7])
def test_backward_in_list_segments_gives_the_same_gradients(dev, monkeypatch, segments):
    """so_step_desc.bwd_seg_len (round 5): the forward rasteriser leaves every pixel's (live transmittance, accumulated colour) at
    the segment boundaries of its tile's list, and the backward runs one workgroup per (tile, segment), each from the state at the
    far end of its segment.  Same image (the forward is untouched), same loss, gradients equal up to the order of float additions
    -- on a cloud gathered in the middle of a small image (lists of 0 ... 3000+ entries: pixels that stop in the first segment,
    pixels that run through all of them, tiles shorter than one segment, lists longer than all segments together)."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 30_000, 256, 160
    out = {}
    for seg in (1, segments):
        monkeypatch.setenv("SPLAT_ONE_AMD_BWD_SEGMENTS", str(seg))
        r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
        with torch.no_grad():
            r.splats["means"].mul_(0.25)
        eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False, fuse_adam=False)
        eng.set_views(c2w, Ks, pixels)
        eng.fwd_bwd()
        torch.cuda.synchronize()
        d = eng._desc()
        assert eng.binned and eng.cfg["raster_impl"] == 0 and (d.bwd_seg_count, d.bwd_seg_len > 0) == ((seg, True) if seg > 1 else (0, False))
        if seg > 1:     # (the entries per segment are sized by the fullest list of the probe; one test with lists longer than all segments)
            if segments == 7:
                eng._list_stats = (eng._list_stats[0] // 3, eng._list_stats[1])
                eng.fwd_bwd()
                torch.cuda.synchronize()
                assert eng._desc().bwd_seg_len * segments < eng._fullest_tile()
        out[seg] = dict(img=eng.ws["render_colors"].clone(), loss=eng.loss().clone(), grads={k: v.grad.detach().clone() for k, v in r.splats.items()})
    a, b = out[1], out[segments]
    assert torch.equal(a["img"], b["img"]) and torch.allclose(a["loss"], b["loss"], rtol=1e-6, atol=0)      # (the loss sums are atomics)
    for k in a["grads"]:
        # (a segment starts from the forward's own transmittance; the one-chain walk re-derives it by a product of reciprocals:
        # measured 1.3e-5 apart with four segments, 3.3e-5 with seven -- two float32 routes to the same number, the oracle's bar is 1e-3)
        assert rel_err(b["grads"][k], a["grads"][k]) < 1e-4, (segments, k, rel_err(b["grads"][k], a["grads"][k]))


def test_tile_tables_kept_per_view_change_nothing_but_the_launches(dev, monkeypatch):
    """The engine keeps the workgroup -> tile table it built for a view and hands it back at the next visits of the same view
    (so_step_desc.tile_order_ready, so_step_inputs order_src): built at the first visit and every `order_refresh`-th one, a valid
    permutation at all times, and training with kept tables gives the images of training with a table per step (an order only
    schedules)."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 20_000, 256, 160
    outs = {}
    for cache in ("1", "0"):
        monkeypatch.setenv("SPLAT_ONE_AMD_ORDER_CACHE", cache)
        r, c2w, Ks, pixels = _make(dev, N, W, H, "ref")
        views = [(c2w, Ks, pixels), (c2w.clone(), Ks, (pixels * 0.5).contiguous())]
        eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=True)
        eng.order_refresh = 3                    # (built at the first visit of a view and every third one)
        modes, imgs = [], []
        for it in range(12):
            c, k, px = views[it % 2]
            eng.set_views(c, k, px, schedule=True)
            modes.append(eng._order_mode)
            eng.step()
            torch.cuda.synchronize()
            order = eng.ws["tile_order"].cpu()
            assert eng._order_mode == "none" or torch.equal(torch.sort(order).values, torch.arange(eng.M, dtype=torch.int32))
            imgs.append(eng.ws["render_colors"].clone())
        outs[cache] = (modes, imgs)
        if cache == "1":
            assert set(modes[2:]) <= {"build", "kept"} and modes.count("build") >= 4 and modes.count("kept") >= 6, modes
            assert len(eng._order_cache) == 2
        else:
            assert "kept" not in modes and "build" not in modes, modes
    # (the first image: same parameters, bit for bit; later ones: two trainings whose gradients are atomic sums)
    assert torch.equal(outs["1"][1][0], outs["0"][1][0])
    for a, b in zip(outs["1"][1], outs["0"][1]):
        assert (a - b).abs().max().item() < 1e-3
