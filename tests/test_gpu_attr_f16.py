"""float16 attribute storage (BASELINE.json configs[4], include/splat_one_amd.h "float16 attribute storage").

The contract: arithmetic stays float32, so every result equals the float32 path evaluated on the half-rounded
attribute values -- which is what the float64 oracle is given here, at the usual bars (forward <= 1e-4 mean L1,
gradients <= 1e-3 relative).  The float32 masters stay authoritative and their halves never diverge from them."""
import ctypes

import pytest
import torch

from oracle import c_oracle as CO
from oracle import ssim_oracle as SSO
from oracle import torch_oracle as O
from tests.test_gpu_engine import _make

pytestmark = pytest.mark.gpu

F16_KEYS = ("quats", "scales", "sh0", "shN")


def _round_masters_(splats):
    with torch.no_grad():
        for k in F16_KEYS:
            splats[k].copy_(splats[k].half().float())


def test_attr_pack_rows_are_the_rounded_masters(dev):
    from splat_one_amd import _lib
    for K in (1, 4, 16):
        N = 3001
        g = torch.Generator().manual_seed(K)
        quats = torch.randn(N, 4, generator=g).to(dev)
        scales = (torch.randn(N, 3, generator=g) * 3).to(dev)
        sh0 = torch.randn(N, 1, 3, generator=g).to(dev)
        shN = (torch.randn(N, K - 1, 3, generator=g) * 0.3).to(dev)
        scales[0, 0] = 1e6                                   # beyond the half range: +inf, like torch .half()
        stride = int(_lib.load().so_attr_rec_stride(K))
        assert stride == 16 + (6 * K + 15) // 16 * 16
        arec = torch.full((N * stride // 4,), float("nan"), device=dev)
        _lib.call("so_attr_pack_f16", N, K, _lib.ptr(scales), _lib.ptr(quats), _lib.ptr(sh0), _lib.ptr(shN) if K > 1 else 0,
                  _lib.ptr(arec), _lib.stream())
        h = arec.view(torch.float16).view(N, stride // 2)
        assert torch.equal(h[:, 0:4], quats.half())
        assert torch.equal(h[:, 4:7], scales.half())
        assert torch.equal(h[:, 8:11], sh0.view(N, 3).half())
        assert torch.equal(h[:, 11:8 + 3 * K], shN.reshape(N, -1).half())
        assert (h[:, 7] == 0).all() and (h[:, 8 + 3 * K:] == 0).all()      # padding is zero, not garbage


@pytest.mark.parametrize("C,kw", [(1, {}), (2, {"antialiased": True, "opacity_reg": 0.01, "scale_reg": 0.01})])
def test_engine_f16_attributes_match_oracle_at_rounded_values(dev, C, kw):
    from splat_one_amd.engine import FusedEngine
    N, W, H = 6000, 160, 96
    r, c2w, Ks, pixels = _make(dev, N, W, H, "ref", C, **kw)
    r.step = 5
    st = r.cfg.strategy.initialize_state(1.0)
    eng = FusedEngine(r.splats, r.optimizers, W, H, C, sh_degree=3, strategy_state=st, use_graph=False, attr_dtype="f16",
                      antialiased=kw.get("antialiased", False), opacity_reg=kw.get("opacity_reg", 0.0),
                      scale_reg=kw.get("scale_reg", 0.0))
    rows = eng.attr_rows()
    for k in F16_KEYS:                                       # the rows hold the rounded masters; masters are untouched
        assert torch.equal(rows[k], r.splats[k].detach().half().float())
    assert not torch.equal(rows["shN"], r.splats["shN"].detach())
    eng.set_views(c2w, Ks, pixels)
    eng.fwd_bwd()
    g_eng = {k: v.grad.detach().clone().cpu().double() for k, v in r.splats.items()}
    img = eng.ws["render_colors"].clone()
    assert eng.stats()["overflow"] == 0
    # float64 oracle at the values the kernels actually read
    p = {k: (v.detach().half().float() if k in F16_KEYS else v.detach()).cpu().clone().requires_grad_(True)
         for k, v in r.splats.items()}
    rc, ra, meta = O.rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                   torch.cat([p["sh0"], p["shN"]], 1), torch.linalg.inv(c2w.cpu()), Ks.cpu(), W, H,
                                   sh_degree=3, near_plane=0.01, far_plane=1e8,
                                   rasterize_mode="antialiased" if kw.get("antialiased") else "classic",
                                   raster_fn=CO.raster_fn())
    loss_o, l1_o, ss_o = SSO.photometric_loss(rc, pixels.cpu(), 0.2)
    if kw.get("opacity_reg", 0) > 0:
        loss_o = loss_o + kw["opacity_reg"] * torch.sigmoid(p["opacities"].double()).abs().mean()
    if kw.get("scale_reg", 0) > 0:
        loss_o = loss_o + kw["scale_reg"] * torch.exp(p["scales"].double()).abs().mean()
    loss_o.backward()
    assert (img.cpu().double() - rc.double()).abs().mean().item() < 1e-4
    for k in g_eng:
        floor = 1e-5 * p["scales"].grad.norm().item() if k == "quats" else 0.0
        err = (g_eng[k] - p[k].grad.double()).norm().item()
        assert err <= 1e-3 * p[k].grad.norm().item() + floor, (k, err, p[k].grad.norm().item())
    # and the float32-storage engine on masters that ARE the rounded values computes the same thing
    r2, _, _, _ = _make(dev, N, W, H, "ref", C, **kw)
    _round_masters_(r2.splats)
    eng2 = FusedEngine(r2.splats, r2.optimizers, W, H, C, sh_degree=3, use_graph=False,
                       antialiased=kw.get("antialiased", False), opacity_reg=kw.get("opacity_reg", 0.0),
                       scale_reg=kw.get("scale_reg", 0.0))
    eng2.set_views(c2w, Ks, pixels)
    eng2.fwd_bwd()
    assert torch.equal(eng2.ws["radii"], eng.ws["radii"])
    assert (eng2.ws["render_colors"] - img).abs().max().item() < 1e-5
    for k, v in r2.splats.items():
        assert (v.grad.cpu().double() - g_eng[k]).norm().item() <= 1e-5 * g_eng[k].norm().item() + 1e-12, k


@pytest.mark.parametrize("use_graph,how", [(False, "fused"), (True, "fused"), (False, "repack"), (True, "repack"), (True, "scatter")])
def test_engine_f16_rows_follow_the_optimiser(dev, use_graph, how):
    """Adam updates the float32 masters exactly as in float32 storage given the same gradients, and the step leaves
    rows == masters.half() bit for bit -- written by the backward kernel that applies the update (the fused optimiser, default),
    by the coalesced re-pack that follows a separate Adam launch (fuse_adam=False), or by the scatter inside that launch
    (f16_repack = False); training makes progress; a rebuild repacks."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 5000, 128, 96                        # 5000 = 78 x 64 + 8: a partial last wave in the row sweep
    r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc")
    r.step = 10
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, lr_gamma_means=r.lr_gamma, use_graph=use_graph,
                      attr_dtype="f16", fuse_adam=(how == "fused"))
    eng.f16_repack = how != "scatter"
    assert eng._fusable(True) == (how == "fused")
    losses = []
    for _ in range(12):
        eng.set_views(c2w, Ks, pixels, schedule=True)
        eng.step()
        losses.append(eng.loss()[0].item())
    assert eng.stats()["overflow"] == 0
    assert losses[-1] < losses[0]
    rows = eng.attr_rows()
    for k in F16_KEYS:
        assert torch.equal(rows[k], r.splats[k].detach().half().float()), k
        assert not torch.equal(rows[k], r.splats[k].detach()), k      # masters keep their float32 precision
    # external edit + rebuild (what densification does)
    with torch.no_grad():
        r.splats["shN"].mul_(0.5)
    assert not torch.equal(eng.attr_rows()["shN"], r.splats["shN"].detach().half().float())
    eng.rebuild()
    assert torch.equal(eng.attr_rows()["shN"], r.splats["shN"].detach().half().float())


@pytest.mark.parametrize("use_graph,device_refine", [(False, False), (True, True)])
def test_engine_f16_fused_optimiser_equals_the_two_kernel_step(dev, use_graph, device_refine):
    """With float16 rows the optimiser fused into the backward (which also re-packs the rows it owns) gives the parameters,
    moments and rows of backward + so_adam_step_dev + so_attr_pack_f16 -- through the SH ramp (columns of shN that have
    no gradient yet still decay their moments and are re-packed), with two views, regularisers and densification statistics;
    the padding halves of every row stay zero."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 5003, 128, 96
    out = {}
    for fuse in (False, True):
        r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc", 2, opacity_reg=0.01, scale_reg=0.01)
        st = r.cfg.strategy.initialize_state(1.0)
        eng = FusedEngine(r.splats, r.optimizers, W, H, 2, sh_degree=0, strategy_state=st, lr_gamma_means=r.lr_gamma,
                          use_graph=use_graph, fuse_adam=fuse, opacity_reg=0.01, scale_reg=0.01, attr_dtype="f16",
                          device_refine=device_refine, capacity=(8192 if device_refine else None))
        for it in range(6):
            eng.set_sh_degree(min(it, 3))
            eng.set_views(c2w, Ks, pixels, schedule=True)
            eng.step()
        torch.cuda.synchronize()
        eng.sync_host() if device_refine else None
        assert eng.stats()["overflow"] == 0 and eng.steps_done == 6
        rows = eng.attr_rows()
        for k in F16_KEYS:
            assert torch.equal(rows[k][:N], eng.splats[k].detach()[:N].half().float()), (fuse, k)
        raw = eng.ws["arec"].view(torch.int16).view(-1, eng.attr_stride // 2)[:N]
        assert int(raw[:, 7].abs().max()) == 0                      # K = 16: 8 + 48 halves fill the 112-byte row exactly
        stats = eng.dstats if device_refine else st
        out[fuse] = ({k: v.detach()[:N].clone() for k, v in eng.splats.items()},
                     {k: (eng.optimizers[k].state[eng.splats[k]]["exp_avg"][:N].clone(),
                          eng.optimizers[k].state[eng.splats[k]]["exp_avg_sq"][:N].clone()) for k in eng.splats.keys()},
                     stats["grad2d"][:N].clone(), stats["count"][:N].clone(), eng.loss().clone())
    (pa, ma, g2a, cna, la), (pb, mb, g2b, cnb, lb) = out[False], out[True]
    rel = lambda a, b: (a - b).norm().item() / (a.norm().item() + 1e-30)
    for k in pa:
        assert rel(pa[k], pb[k]) < 2e-5, k
        assert rel(ma[k][0], mb[k][0]) < 1e-4 and rel(ma[k][1], mb[k][1]) < 1e-4, k
    assert rel(g2a, g2b) < 1e-4 and torch.equal(cna, cnb) and (la - lb).abs().max().item() < 1e-5


def test_engine_f16_sh_degree_ramp_and_degree0_only(dev):
    """K = 1 (no shN at all) and the SH ramp 0 -> 3 on K = 16 rows."""
    from splat_one_amd.engine import FusedEngine
    N, W, H = 3000, 96, 64
    r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc", sh_degree=0)
    assert r.splats["shN"].shape[1] == 0
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=0, use_graph=False, attr_dtype="f16")
    assert eng.attr_stride == 32
    eng.set_views(c2w, Ks, pixels, schedule=True)
    eng.step()
    assert torch.isfinite(eng.loss()).all()
    assert torch.equal(eng.attr_rows()["sh0"], r.splats["sh0"].detach().half().float())
    r, c2w, Ks, pixels = _make(dev, N, W, H, "mcmc")
    e16 = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=0, use_graph=True, attr_dtype="f16")
    r2, _, _, _ = _make(dev, N, W, H, "mcmc")
    _round_masters_(r2.splats)
    e32 = FusedEngine(r2.splats, r2.optimizers, W, H, 1, sh_degree=0, use_graph=True)
    for deg in (0, 1, 2, 3):
        for e in (e16, e32):
            e.set_sh_degree(deg)
            e.set_views(c2w, Ks, pixels)
            e.fwd_bwd()
        assert (e16.ws["render_colors"] - e32.ws["render_colors"]).abs().max().item() < 1e-5, deg
        for k in r.splats:
            a, b = r.splats[k].grad, r2.splats[k].grad
            assert (a - b).norm().item() <= 1e-5 * b.norm().item() + 1e-12, (deg, k)


def test_f16_entry_points_reject_bad_rows(dev):
    from splat_one_amd import _lib
    z = torch.zeros(64, device=dev)
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        _lib.call("so_attr_pack_f16", 1, 16, _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), z.data_ptr() + 4, _lib.stream())
    with pytest.raises(RuntimeError, match="arec"):
        _lib.call("so_preprocess_fwd_f16", 1, 1, 16, 3, _lib.ptr(z), _lib.ptr(z), 0, _lib.ptr(z), _lib.ptr(z), 16, 16,
                  0.3, 0.01, 100.0, 0.0, 0, 0, 16, *([_lib.ptr(z)] * 7), 0, 0, 0, 0, 0, 0, 0, 0, 0, _lib.stream())
