"""Packed mode (`packed=True`, gsplat's default; `Config.packed` / `Config.sparse_grad` at gsplat_trainer.py:132-135,
487-489, 705-724, 751): one row per (camera, Gaussian) pair with a positive radius.  The packed call must give the
images, gradients, tile lists and densification statistics of the dense [C,N] call on the same inputs."""
import math

import pytest
import torch

from splat_one_amd.scene import make_scene, pinhole_K, ring_cameras
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _scene(dev, C, N=3000, W=128, H=96):
    splats, c2w, Ks = make_scene(N, W, H, "ref", n_views=C)
    g = torch.Generator().manual_seed(5)
    splats = {k: v.detach().clone() for k, v in splats.items()}
    splats["scales"] = splats["scales"] + torch.randn(N, 3, generator=g) * 0.4      # anisotropic
    splats["means"][: N // 10, 2] -= 40.0                                           # a tenth behind every camera
    viewmats = torch.linalg.inv(c2w)
    return splats, viewmats.to(dev), Ks.to(dev), W, H


def _render(dev, splats, viewmats, Ks, W, H, **kw):
    from splat_one_amd import rasterization
    C = viewmats.shape[0]
    p = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in splats.items()}
    colors = torch.cat([p["sh0"], p["shN"]], 1)
    rc, ra, meta = rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), colors,
                                 viewmats, Ks, W, H, sh_degree=3, near_plane=0.01, far_plane=1e8, **kw)
    meta["means2d"].retain_grad()
    g = torch.Generator().manual_seed(11)
    w_rgb = torch.rand(C, H, W, rc.shape[-1], generator=g).to(dev)
    w_a = torch.rand(C, H, W, 1, generator=g).to(dev)
    ((rc * w_rgb).sum() + (ra * w_a).sum()).backward()
    return rc.detach(), ra.detach(), p, meta


@pytest.mark.parametrize("C,kw", [(1, {}), (2, {}), (3, dict(rasterize_mode="antialiased", absgrad=True, render_mode="RGB+ED")),
                                  (2, dict(camera_model="fisheye", backgrounds=torch.tensor([[0.2, 0.4, 0.6]] * 2)))])
def test_packed_rasterization_equals_dense(dev, C, kw):
    kw = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    splats, viewmats, Ks, W, H = _scene(dev, C)
    N = splats["means"].shape[0]
    # (fused=False: the dense call through the same operator composition as the packed one -- this test is about the two
    # LAYOUTS; the one-call path of the dense common shape is held against it in tests/test_gpu_raster_op.py)
    rc_d, ra_d, p_d, m_d = _render(dev, splats, viewmats, Ks, W, H, packed=False, fused=False, **kw)
    rc_p, ra_p, p_p, m_p = _render(dev, splats, viewmats, Ks, W, H, packed=True, **kw)
    # layout: the pairs with a positive radius, camera-major
    cam, gid = torch.nonzero(m_d["radii"] > 0, as_tuple=True)
    assert 0 < cam.numel() < C * N
    assert torch.equal(m_p["camera_ids"], cam) and torch.equal(m_p["gaussian_ids"], gid)
    for k in ("radii", "means2d", "depths", "conics", "opacities"):
        assert m_p[k].shape[0] == cam.numel() and torch.equal(m_p[k], m_d[k][cam, gid]), k
    # same tile lists (flatten_ids name packed rows), same images (see below)
    assert torch.equal(m_p["isect_offsets"], m_d["isect_offsets"])
    f = m_p["flatten_ids"].long()
    assert torch.equal(cam[f] * N + gid[f], m_d["flatten_ids"].long())
    assert torch.equal(m_p["isect_ids"], m_d["isect_ids"])
    assert torch.equal(m_p["tiles_per_gauss"], m_d["tiles_per_gauss"][cam, gid])
    # alphas bit for bit; colours to rounding: the dense call evaluates view directions + SH + clamp in one fused launch,
    # the packed one through the separate operators (two compilations of the same arithmetic)
    assert torch.equal(ra_p, ra_d) and (rc_p - rc_d).abs().max().item() <= 2e-6
    # gradients (float atomics: order varies)
    for k in p_d:
        assert rel_err(p_p[k].grad.cpu().double(), p_d[k].grad.cpu().double()) < 1e-5, k
    assert rel_err(m_p["means2d"].grad.cpu().double(), m_d["means2d"].grad[cam, gid].cpu().double()) < 1e-5
    if kw.get("absgrad"):
        a_p, a_d = m_p["means2d"].absgrad, m_d["means2d"].absgrad[cam, gid]
        assert rel_err(a_p.cpu().double(), a_d.cpu().double()) < 1e-5
        assert (a_p + 1e-12 >= m_p["means2d"].grad.abs()).all()


def test_packed_nothing_visible(dev):
    from splat_one_amd import rasterization
    splats, viewmats, Ks, W, H = _scene(dev, 2, N=64)
    p = {k: v.to(dev) for k, v in splats.items()}
    means = p["means"].clone()
    means[:, 1] += 1.0e4                                           # far above every camera's frustum
    bg = torch.tensor([[0.1, 0.2, 0.3]] * 2, device=dev)
    rc, ra, meta = rasterization(means, p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                 torch.cat([p["sh0"], p["shN"]], 1), viewmats, Ks, W, H, sh_degree=3, backgrounds=bg)
    assert meta["gaussian_ids"].numel() == 0 and meta["means2d"].shape == (0, 2) and meta["flatten_ids"].numel() == 0
    assert torch.equal(ra, torch.zeros_like(ra)) and torch.allclose(rc, bg[:, None, None, :].expand_as(rc))


def test_sparse_grad(dev):
    splats, viewmats, Ks, W, H = _scene(dev, 2)
    _, _, p_d, m_d = _render(dev, splats, viewmats, Ks, W, H, packed=False)
    _, _, p_s, m_s = _render(dev, splats, viewmats, Ks, W, H, packed=True, sparse_grad=True)
    visible = torch.unique(m_s["gaussian_ids"])
    for k in ("quats", "scales"):
        g = p_s[k].grad
        assert g.is_sparse and torch.equal(g.coalesce().indices()[0], visible), k      # rows of the visible Gaussians only
    for k in p_d:
        g = p_s[k].grad
        g = g.to_dense() if g.is_sparse else g
        assert rel_err(g.cpu().double(), p_d[k].grad.cpu().double()) < 1e-5, k
    from splat_one_amd import rasterization
    with pytest.raises(AssertionError):
        rasterization(*[v.to(dev) for v in (splats["means"], splats["quats"], splats["scales"].exp(), splats["opacities"].sigmoid(),
                                            splats["sh0"][:, 0])], viewmats, Ks, W, H, packed=False, sparse_grad=True)


def _runner(dev, N=3000, **kw):
    from splat_one_amd.strategy import DefaultStrategy
    from splat_one_amd.trainer import Config, Runner
    strat = DefaultStrategy(refine_start_iter=2, refine_every=3, reset_every=50, refine_stop_iter=1000, grow_grad2d=5e-5,
                            refine_scale2d_stop_iter=100, verbose=False)
    cfg = Config(init_num_pts=N, init_scale=0.3, init_opa=0.3, shN_init_std=0.05, sh_degree_interval=1, max_steps=200,
                 strategy=strat, fused=False, **kw)
    r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
    with torch.no_grad():   # anisotropic scales: otherwise the quaternion gradient is pure rounding noise (Adam: +-lr steps)
        r.splats["scales"].add_((torch.randn(N, 3, generator=torch.Generator().manual_seed(7)) * 0.4).to(dev))
    return r


@pytest.mark.parametrize("visible_adam", [False, True])
def test_trainer_packed_equals_dense(dev, visible_adam):
    """Config.packed through the training step: densification statistics from info["gaussian_ids"] (strategy
    `packed=True`), SelectiveAdam's visibility mask by scatter (:719-724) -- same Gaussians after refinements."""
    W, H, B = 128, 96, 2
    c2w = ring_cameras(8)[:B].to(dev)
    Ks = pinhole_K(W, H)[None].repeat(B, 1, 1).to(dev)
    pixels = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    out = {}
    for packed in (False, True):
        r = _runner(dev, packed=packed, visible_adam=visible_adam, batch_size=B)
        stats = []
        for step in range(5):
            r.train_step(c2w, Ks, pixels)
            if step == 1:
                stats = [r.strategy_state[k].clone() for k in ("grad2d", "count", "radii")]
        out[packed] = (r, stats)
    (rd, sd), (rp, sp) = out[False], out[True]
    assert rp.last_info["gaussian_ids"] is not None and rd.last_info["gaussian_ids"] is None
    for a, b, k in zip(sp, sd, ("grad2d", "count", "radii")):
        assert rel_err(a.cpu().double(), b.cpu().double()) < 1e-4, k
    n = len(rd.splats["means"])
    assert n != 3000 and len(rp.splats["means"]) == n             # a refinement ran, and took the same decisions
    for k in rd.splats.keys():
        # five Adam steps: where a gradient is rounding noise of the float atomics, Adam's normalisation turns a flipped
        # sign into a full learning-rate step (the two-rank tests of test_gpu_trainer.py allow 2e-4 after four)
        assert rel_err(rp.splats[k].detach().cpu().double(), rd.splats[k].detach().cpu().double()) < 5e-4, k


def test_trainer_sparse_grad(dev):
    """Config.sparse_grad (:267-268, 705-717): torch.optim.SparseAdam on row-sparse gradients -- Gaussians no camera sees
    keep their parameters and moments untouched."""
    W, H = 128, 96
    from splat_one_amd.strategy import DefaultStrategy
    from splat_one_amd.trainer import Config, Runner
    cfg = Config(init_num_pts=3000, init_scale=0.3, init_opa=0.3, shN_init_std=0.05, sh_degree_interval=1, fused=False,
                 packed=True, sparse_grad=True, strategy=DefaultStrategy(verbose=False))
    r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
    assert all(isinstance(o, torch.optim.SparseAdam) for o in r.optimizers.values())
    with torch.no_grad():
        r.splats["means"][:300, 2] -= 40.0                         # never visible
    before = {k: v.detach().clone() for k, v in r.splats.items()}
    c2w = ring_cameras(8)[:1].to(dev)
    Ks = pinhole_K(W, H)[None].to(dev)
    pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    losses = [float(r.train_step(c2w, Ks, pixels)) for _ in range(4)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]
    vis = torch.zeros(3000, dtype=torch.bool, device=dev)
    vis[r.last_info["gaussian_ids"]] = True
    assert not vis[:300].any() and vis.sum() > 1000
    for k, v in r.splats.items():
        assert torch.equal(v.detach()[:300], before[k][:300]), k
        assert not torch.equal(v.detach()[vis], before[k][vis]), k


@pytest.mark.parametrize("C,sparse_grad,camera_model", [(1, True, "pinhole"), (3, True, "pinhole"), (2, False, "fisheye")])
def test_packed_against_the_oracle(dev, C, sparse_grad, camera_model):
    """Oracle parity of the packed layout itself (not packed-vs-dense of the same kernels): rows, ids, image and every
    gradient -- row-sparse COO ones densified -- against the float64 oracle's dense [C,N] results gathered at
    (camera_ids, gaussian_ids).  No [C,N] array is allocated by the product on this path (so_projection_packed)."""
    from oracle import c_oracle as CO
    from oracle import torch_oracle as O
    from splat_one_amd import rasterization
    splats, viewmats, Ks, W, H = _scene(dev, C)
    N = splats["means"].shape[0]
    g = torch.Generator().manual_seed(11)
    w_rgb = torch.rand(C, H, W, 3, generator=g)
    w_a = torch.rand(C, H, W, 1, generator=g)
    # product, packed
    p = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in splats.items()}
    torch.cuda.reset_peak_memory_stats()
    rc, ra, meta = rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                 torch.cat([p["sh0"], p["shN"]], 1), viewmats, Ks, W, H, sh_degree=3, near_plane=0.01,
                                 far_plane=1e8, packed=True, sparse_grad=sparse_grad, camera_model=camera_model)
    meta["means2d"].retain_grad()
    ((rc * w_rgb.to(dev)).sum() + (ra * w_a.to(dev)).sum()).backward()
    # oracle, dense
    q = {k: v.detach().clone().requires_grad_(True) for k, v in splats.items()}
    rc_o, ra_o, m_o = O.rasterization(q["means"], q["quats"], torch.exp(q["scales"]), torch.sigmoid(q["opacities"]),
                                      torch.cat([q["sh0"], q["shN"]], 1), viewmats.cpu(), Ks.cpu(), W, H, sh_degree=3,
                                      near_plane=0.01, far_plane=1e8, camera_model=camera_model, raster_fn=CO.raster_fn())
    m_o["means2d"].retain_grad()
    ((rc_o * w_rgb).sum() + (ra_o * w_a).sum()).backward()
    cam, gid = torch.nonzero(m_o["radii"] > 0, as_tuple=True)
    assert torch.equal(meta["camera_ids"].cpu(), cam) and torch.equal(meta["gaussian_ids"].cpu(), gid)   # sorted, camera-major
    assert torch.equal(meta["radii"].cpu(), m_o["radii"][cam, gid])
    assert rel_err(meta["means2d"], m_o["means2d"][cam, gid]) < 1e-5 and rel_err(meta["conics"], m_o["conics"][cam, gid]) < 1e-4
    assert rel_err(meta["depths"], m_o["depths"][cam, gid]) < 1e-6
    assert (rc.detach().cpu().double() - rc_o).abs().mean().item() <= 1e-4
    assert rel_err(meta["means2d"].grad, m_o["means2d"].grad[cam, gid]) < 1e-3
    for k in q:
        gk = p[k].grad
        if sparse_grad and k in ("quats", "scales"):              # (means also receives a dense part through the SH view directions)
            assert gk.is_sparse
            rows = gk._indices()[0].cpu()       # (autograd's exp / accumulate steps do not keep the `coalesced` flag; the rows do)
            assert torch.equal(rows, torch.unique(gid))             # exactly the visible Gaussians, each once, sorted
        if gk.is_sparse:
            gk = gk.to_dense()
        floor = 1e-5 * q["scales"].grad.norm().item() if k == "quats" else 0.0
        err = (gk.detach().cpu().double() - q[k].grad.double()).norm().item()
        assert err <= 1e-3 * q[k].grad.norm().item() + floor, (k, err)


@pytest.mark.parametrize("seed", list(range(12)))
def test_packed_projection_random_sizes(dev, seed):
    """`fully_fused_projection(packed=True)` (count / scan / write passes over 1024-pair blocks) against the dense call on
    sizes around the block, workgroup and wave boundaries, several cameras, all camera models: the same pairs in the
    same (camera-major) order with the same values."""
    import random
    from splat_one_amd.ops import fully_fused_projection
    rnd = random.Random(40 + seed)
    C = rnd.choice([1, 2, 3, 5])
    N = rnd.choice([1, 63, 64, 65, 255, 1023, 1024, 1025, 2047, 2049, rnd.randint(2, 6000), rnd.randint(2, 6000)])
    model = rnd.choice(["pinhole", "fisheye", "ortho", "spherical"])
    W, H = rnd.randint(40, 300), rnd.randint(30, 200)
    g = torch.Generator().manual_seed(500 + seed)
    means = ((torch.rand(N, 3, generator=g) * 2 - 1) * 3).to(dev)
    quats = torch.randn(N, 4, generator=g).to(dev)
    scales = (torch.rand(N, 3, generator=g) * 0.3 + 0.02).to(dev)
    from splat_one_amd.scene import lookat_c2w
    c2w = torch.stack([lookat_c2w((7.0 * math.sin(1.3 * i + seed), 0.4 * i - 0.6, -7.0 * math.cos(1.3 * i + seed))) for i in range(C)])
    viewmats = torch.linalg.inv(c2w).to(dev)
    f = (0.05 if model == "ortho" else 0.9) * max(W, H)
    Ks = torch.tensor([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]])[None].repeat(C, 1, 1).to(dev)
    kw = dict(eps2d=0.3, near_plane=0.01, far_plane=1e8, radius_clip=rnd.choice([0.0, 0.0, 2.0]), camera_model=model,
              calc_compensations=rnd.random() < 0.5)
    radii, m2, dep, con, comp = fully_fused_projection(means, None, quats, scales, viewmats, Ks, W, H, packed=False, **kw)
    cam, gid, radii_p, m2_p, dep_p, con_p, comp_p = fully_fused_projection(means, None, quats, scales, viewmats, Ks, W, H, packed=True, **kw)
    wc, wg = torch.nonzero(radii > 0, as_tuple=True)
    assert torch.equal(cam, wc) and torch.equal(gid, wg), (C, N, model)
    assert torch.equal(radii_p, radii[wc, wg])
    # values: two compilations of the same arithmetic (the packed kernel and the dense one), equal to rounding
    close = lambda a, b: bool(((a - b).abs() <= 1e-4 * b.abs() + 1e-4).all())    # (pixels / conic entries: a few ulp)
    assert close(m2_p, m2[wc, wg]) and close(dep_p, dep[wc, wg]) and close(con_p, con[wc, wg])
    if comp is not None:
        assert close(comp_p, comp[wc, wg])
