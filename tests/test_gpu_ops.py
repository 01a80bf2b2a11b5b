"""GPU parity of every operator of the path against the float64 oracle, on identical inputs.
Bars (BASELINE.json north_star): forward <= 1e-4 per-pixel L1, gradients <= 1e-3 relative
(||g-g*||/||g*|| per tensor); integer / index outputs bit-exact."""
import math

import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as O
from tests.util import rel_err, small_scene, two_cameras

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
GRAD_TOL = 1e-3


@pytest.mark.parametrize("camera_model", ["pinhole", "ortho", "fisheye", "spherical"])
@pytest.mark.parametrize("use_covars", [False, True])
def test_projection_fwd_bwd(dev, camera_model, use_covars):
    from splat_one_amd.ops import fully_fused_projection
    W, H, N = 96, 64, 3000
    means, quats, scales, _, _ = small_scene(N)
    viewmats, Ks = two_cameras(W, H)
    if camera_model == "ortho":
        Ks[:, 0, 0], Ks[:, 1, 1] = 12.0, 11.0
    C = viewmats.shape[0]
    g = torch.Generator().manual_seed(3)
    wm, wd, wc, wp = (torch.randn(C, N, 2, generator=g), torch.randn(C, N, generator=g),
                      torch.randn(C, N, 3, generator=g), torch.randn(C, N, generator=g))

    def run(fn, to, dt, keep=None):
        m = means.detach().clone().to(to).requires_grad_()
        V = viewmats.detach().clone().to(to).requires_grad_()
        if use_covars:
            cov = O.quat_scale_to_covar(quats.double(), scales.double()).to(dt).to(to).requires_grad_()
            q = s = None
        else:
            cov = None
            q, s = quats.detach().clone().to(to).requires_grad_(), scales.detach().clone().to(to).requires_grad_()
        radii, m2, dep, con, comp = fn(m, cov, q, s, V, Ks.to(to), W, H, eps2d=0.3, near_plane=0.01,
                                       far_plane=100.0, calc_compensations=True, camera_model=camera_model)
        k = torch.ones(C, N) if keep is None else keep
        k = k.to(m2)
        loss = ((m2 * (wm.to(m2) * k[..., None])).sum() + (dep * (wd.to(dep) * k)).sum()
                + (con * (wc.to(con) * k[..., None])).sum() + (comp * (wp.to(comp) * k)).sum())
        loss.backward()
        grads = {"means": m.grad, "viewmats": V.grad[:, :3, :]}
        if use_covars:
            grads["covars"] = cov.grad + cov.grad.transpose(-1, -2)   # symmetric part is what is defined
        else:
            grads.update(quats=q.grad, scales=s.grad)
        return radii, m2, dep, con, comp, grads

    r_h, m2_h, d_h, c_h, p_h, _ = run(fully_fused_projection, dev, torch.float32)
    r_o, m2_o, d_o, c_o, p_o, _ = run(O.fully_fused_projection, "cpu", torch.float32)
    r_h = r_h.cpu()
    # radii: integer; fp32 vs fp64 may differ by one on ceil() boundaries / culling borderline
    same = (r_h == r_o)
    assert same.float().mean() > 0.999, same.float().mean()
    assert (r_h - r_o).abs()[(r_h > 0) & (r_o > 0)].max() <= 1
    vis = ((r_h > 0) & (r_o > 0))
    assert vis.sum() > 500
    assert (m2_h.cpu().double() - m2_o)[vis].abs().max() < 2e-3          # pixels
    assert rel_err(d_h.cpu()[vis], d_o[vis]) < 1e-5
    assert rel_err(c_h.cpu()[vis], c_o[vis]) < 1e-4
    assert (p_h.cpu().double() - p_o)[vis].abs().max() < 1e-4
    # gradients: ALWAYS compared -- over the (camera, Gaussian) pairs both sides project; a pair that one
    # side culls and the other keeps (a radius one ulp from the cull threshold) gets zero upstream weight on both sides,
    # a radius that differs by one pixel does not enter any gradient
    keep = ((r_h > 0) == (r_o > 0)).float()
    assert keep.mean() > 0.999
    _, _, _, _, _, g_h = run(fully_fused_projection, dev, torch.float32, keep)
    _, _, _, _, _, g_o = run(O.fully_fused_projection, "cpu", torch.float32, keep)
    for k in g_o:
        assert g_o[k].abs().max() > 0, k
        assert rel_err(g_h[k], g_o[k]) < GRAD_TOL, (k, rel_err(g_h[k], g_o[k]))


@pytest.mark.parametrize("degree", [0, 1, 2, 3, 4])
def test_spherical_harmonics(dev, degree):
    from splat_one_amd.ops import spherical_harmonics
    N, C, K = 4000, 3, 25
    g = torch.Generator().manual_seed(degree)
    dirs = torch.randn(C, N, 3, generator=g) * 3
    coeffs = torch.randn(N, K, 3, generator=g)
    masks = torch.rand(C, N, generator=g) > 0.2
    w = torch.randn(C, N, 3, generator=g)

    def run(fn, to):
        d = dirs.to(to).requires_grad_()
        c = coeffs.to(to).requires_grad_()
        out = fn(degree, d, c[None].expand(C, -1, -1, -1), masks=masks.to(to))
        (out * w.to(out)).sum().backward()
        return out, d.grad, c.grad

    o_h, vd_h, vc_h = run(spherical_harmonics, dev)
    o_o, vd_o, vc_o = run(O.spherical_harmonics, "cpu")
    assert (o_h.cpu().double() - o_o).abs().max() < 2e-5
    assert rel_err(vc_h, vc_o) < 1e-5
    if degree > 0:
        assert rel_err(vd_h, vd_o) < 1e-4
    nb = (degree + 1) ** 2
    if nb < K:
        assert vc_h[:, nb:].abs().max().item() == 0.0      # bands above the degree: zero gradient
    assert o_h[~masks.to(dev)].abs().max().item() == 0.0   # masked out


def _projected_inputs(N, W, H, scale, seed=1):
    means, quats, scales, opac, sh = small_scene(N, seed=seed, scale=scale, K=1)
    viewmats, Ks = two_cameras(W, H)
    radii, m2, dep, con, _ = O.fully_fused_projection(means, None, quats, scales, viewmats, Ks, W, H,
                                                      near_plane=0.01, far_plane=100.0)
    C = viewmats.shape[0]
    g = torch.Generator().manual_seed(seed + 100)
    colors = torch.rand(C, N, 3, generator=g)
    opacs = opac[None].expand(C, N).contiguous()
    return radii, m2.float(), dep.float(), con.float(), colors, opacs


@pytest.mark.parametrize("tile_size,W,H,scale", [(16, 200, 120, 0.25), (16, 64, 64, 1.5), (8, 100, 60, 0.1)])
def test_isect_bit_exact(dev, tile_size, W, H, scale):
    from splat_one_amd.ops import isect_offset_encode, isect_tiles
    radii, m2, dep, con, _, _ = _projected_inputs(4000, W, H, scale)
    C = radii.shape[0]
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    tpg_o, ids_o, flat_o = O.isect_tiles(m2, radii, dep, tile_size, tw, th)
    off_o = O.isect_offset_encode(ids_o, C, tw, th)
    tpg_h, ids_h, flat_h, off_fill = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), tile_size, tw, th,
                                                 return_offsets=True)
    assert ids_o.numel() > 1000
    assert torch.equal(tpg_h.cpu(), tpg_o)
    assert torch.equal(ids_h.cpu(), ids_o)
    assert torch.equal(flat_h.cpu(), flat_o)
    assert torch.equal(off_fill.cpu(), off_o)
    off_h = isect_offset_encode(ids_h, C, tw, th)
    assert torch.equal(off_h.cpu(), off_o)
    # unsorted emission order (Gaussian-major, row-major tiles)
    _, ids_u_o, flat_u_o = O.isect_tiles(m2, radii, dep, tile_size, tw, th, sort=False)
    _, ids_u_h, flat_u_h = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), tile_size, tw, th, sort=False)
    assert torch.equal(ids_u_h.cpu(), ids_u_o) and torch.equal(flat_u_h.cpu(), flat_u_o)


def test_isect_empty_and_ties(dev):
    from splat_one_amd.ops import isect_tiles
    # nothing visible
    C, N = 2, 50
    z = torch.zeros(C, N, device=dev)
    tpg, ids, flat, off = isect_tiles(torch.zeros(C, N, 2, device=dev), torch.zeros(C, N, dtype=torch.int32, device=dev),
                                      z, 16, 4, 4, return_offsets=True)
    assert ids.numel() == 0 and flat.numel() == 0 and tpg.sum().item() == 0 and off.abs().sum().item() == 0
    # equal depths: ties resolve by ascending flatten id (stable sort of Gaussian-major emission)
    m2 = torch.full((1, 300, 2), 20.0)
    radii = torch.full((1, 300), 5, dtype=torch.int32)
    dep = torch.full((1, 300), 2.5)
    _, ids_o, flat_o = O.isect_tiles(m2, radii, dep, 16, 4, 4)
    _, ids_h, flat_h = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), 16, 4, 4)
    assert torch.equal(flat_h.cpu(), flat_o) and torch.equal(ids_h.cpu(), ids_o)


def test_isect_every_sort_path_boundary(dev):
    """One tile, list lengths around every switch of the per-tile sort (1 / 2 / 4 keys per lane in registers,
    LDS network, long-list kernel), with ties: bit-exact against the oracle's global sort."""
    from splat_one_amd.ops import isect_tiles
    g = torch.Generator().manual_seed(1)
    for N in (1, 2, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 300, 511, 512, 513, 2047, 2048, 2049):
        m2 = torch.rand(1, N, 2, generator=g) * 8 + 4
        radii = torch.ones(1, N, dtype=torch.int32)
        dep = torch.rand(1, N, generator=g) + 0.5
        dep[0, ::5] = 0.75
        _, ids_o, flat_o = O.isect_tiles(m2, radii, dep, 16, 1, 1)
        _, ids_h, flat_h = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), 16, 1, 1)
        assert torch.equal(flat_h.cpu(), flat_o), N
        assert torch.equal(ids_h.cpu(), ids_o), N


def test_bin_counter_placement_device_equals_host(dev):
    """The device evaluates (run * 1031) mod runs with a float32 quotient estimate and an exact correction
    (so_common.hpp bin_counter_index); the host with integer arithmetic (so_bin_counter_index).  Equal for every tile of
    every grid tried -- tiny, odd, the 1080p / 1440p / 4K grids, several views, a run count that 1031 divides, the largest
    the spread applies to."""
    from splat_one_amd import _lib
    f = _lib.load().so_bin_counter_index
    for M in (1, 2, 3, 7, 64, 77, 510, 2062, 8160, 8161, 14400, 32400, 8 * 8160, 2 * 1031 * 7, (1 << 22) - 2, (1 << 22) + 6):
        out = torch.empty(M, dtype=torch.int64, device=dev)
        _lib.call("so_debug_bin_counter_index", M, _lib.ptr(out), _lib.stream())
        got = out.cpu()
        step = max(1, M // 50_000)                                  # (the host side is a Python loop: sample the largest grids)
        ts = list(range(0, M, step)) + [M - 1]
        want = torch.tensor([f(t, M) for t in ts])
        assert torch.equal(got[ts], want), M
        assert torch.equal(torch.sort(got).values, torch.arange(M)), M      # a bijection on the device too


def test_isect_long_lists(dev):
    """Lists longer than the small (2048) and the large (16384) LDS sort capacities: one section of the long-list kernel,
    exactly one, one key more, two and three sections (sorted in LDS section by section, merged by rank)."""
    from splat_one_amd.ops import isect_tiles
    g = torch.Generator().manual_seed(0)
    for N in (3000, 16384, 16385, 20000, 40000):
        m2 = torch.rand(1, N, 2, generator=g) * 8 + 4          # all inside tile (0,0) of a 16x16 image
        radii = torch.ones(1, N, dtype=torch.int32)
        dep = torch.rand(1, N, generator=g) + 0.5
        dep[0, ::7] = 1.0                                      # plenty of ties
        _, ids_o, flat_o = O.isect_tiles(m2, radii, dep, 16, 1, 1)
        _, ids_h, flat_h = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), 16, 1, 1)
        assert ids_o.numel() == N
        assert torch.equal(flat_h.cpu(), flat_o), N
        assert torch.equal(ids_h.cpu(), ids_o), N


@pytest.mark.parametrize("dist", ["uniform", "all_equal", "two_values", "spike", "heavy_tail", "tiny_range"])
def test_isect_long_lists_by_depth_buckets(dev, dist):
    """The long-list kernel deals a section's keys into depth buckets, sorts each bucket in registers and falls back to a
    network for a bucket of more than 256 keys: bit-identical lists whatever the depths look like -- evenly spread, ALL equal
    (one bucket holds the whole section: the order is by Gaussian id alone), two values, a spike of equal depths inside a
    spread, a heavy tail (most buckets empty), a range of a few ulps -- at sizes on both sides of the bucket-count steps
    (4096 / 8192), of one section (16384) and with a ragged second section."""
    from splat_one_amd.ops import isect_tiles
    g = torch.Generator().manual_seed(11)
    for N in (2049, 4096, 4097, 8192, 8200, 16384, 21000):
        m2 = torch.rand(1, N, 2, generator=g) * 8 + 4          # all inside tile (0,0) of a 16x16 image
        radii = torch.ones(1, N, dtype=torch.int32)
        u = torch.rand(1, N, generator=g)
        if dist == "uniform":
            dep = 0.5 + 7.0 * u
        elif dist == "all_equal":
            dep = torch.full((1, N), 3.25)
        elif dist == "two_values":
            dep = torch.where(u < 0.3, torch.tensor(2.0), torch.tensor(9.5))
        elif dist == "spike":
            dep = torch.where(u < 0.4, torch.tensor(4.0), 0.5 + 7.0 * torch.rand(1, N, generator=g))
        elif dist == "heavy_tail":
            dep = 0.2 + torch.exp(12.0 * u * u * u)
        else:
            dep = 5.0 + (torch.randint(0, 9, (1, N), generator=g).float() * 4.76837158203125e-07)   # 9 neighbouring float32 values
        _, ids_o, flat_o = O.isect_tiles(m2, radii, dep, 16, 1, 1)
        _, ids_h, flat_h = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), 16, 1, 1)
        assert ids_o.numel() == N
        assert torch.equal(flat_h.cpu(), flat_o), (dist, N)
        assert torch.equal(ids_h.cpu(), ids_o), (dist, N)


@pytest.mark.parametrize("tile_size,D,with_bg,W,H,scale", [
    (16, 3, False, 200, 120, 0.25), (16, 3, True, 70, 50, 1.0), (16, 4, True, 64, 64, 0.5),
    (8, 3, False, 100, 60, 0.2), (16, 1, False, 64, 48, 0.5), (16, 7, True, 48, 48, 0.5)])
def test_rasterize_fwd_bwd(dev, tile_size, D, with_bg, W, H, scale):
    from splat_one_amd.ops import rasterize_to_pixels
    N = 4000
    radii, m2, dep, con, colors, opacs = _projected_inputs(N, W, H, scale)
    C = radii.shape[0]
    g = torch.Generator().manual_seed(9)
    colors = torch.rand(C, N, D, generator=g)
    bg = torch.rand(C, D, generator=g) if with_bg else None
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    _, ids, flat = O.isect_tiles(m2, radii, dep, tile_size, tw, th)
    off = O.isect_offset_encode(ids, C, tw, th)
    w_c = torch.rand(C, H, W, D, generator=g)
    w_a = torch.rand(C, H, W, 1, generator=g)

    def run(host):
        to = "cpu" if host else dev
        dt = torch.float64 if host else torch.float32
        ins = [t.to(dt).to(to).requires_grad_() for t in (m2, con, colors, opacs)]
        b = None if bg is None else bg.to(dt).to(to).requires_grad_()
        if host:
            absout = []
            rc, ra = CO.rasterize_to_pixels(*ins, W, H, tile_size, off, flat, backgrounds=b, absgrad_out=absout)
        else:
            rc, ra = rasterize_to_pixels(*ins, W, H, tile_size, off.to(dev), flat.to(dev), backgrounds=b, absgrad=True)
        ((rc * w_c.to(rc)).sum() + (ra * w_a.to(ra)).sum()).backward()
        ab = absout[0] if host else ins[0].absgrad
        return rc, ra, [t.grad for t in ins] + [ab] + ([b.grad] if b is not None else [])

    rc_h, ra_h, g_h = run(False)
    rc_o, ra_o, g_o = run(True)
    assert (rc_h.cpu().double() - rc_o).abs().mean().item() < FWD_TOL
    assert (rc_h.cpu().double() - rc_o).abs().max().item() < 5e-3
    assert (ra_h.cpu().double() - ra_o).abs().mean().item() < FWD_TOL
    names = ["means2d", "conics", "colors", "opacities", "absgrad", "backgrounds"]
    for n, a, b in zip(names, g_h, g_o):
        assert rel_err(a, b) < GRAD_TOL, (n, rel_err(a, b))
    assert (g_h[4] * (1 + 1e-4) + 1e-9 >= g_h[0].abs()).all()   # KAT-10: absgrad >= |grad| (up to fp32 summation order)


def test_rasterize_tile_masks_and_static_mode(dev):
    from splat_one_amd.ops import isect_tiles_static, rasterize_to_pixels
    W, H, ts, N = 96, 80, 16, 3000
    radii, m2, dep, con, colors, opacs = _projected_inputs(N, W, H, 0.4)
    C = radii.shape[0]
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    _, ids, flat = O.isect_tiles(m2, radii, dep, ts, tw, th)
    off = O.isect_offset_encode(ids, C, tw, th)
    ins = [t.to(dev) for t in (m2, con, colors, opacs)]
    ref_c, ref_a = rasterize_to_pixels(*ins, W, H, ts, off.to(dev), flat.to(dev))
    # static-capacity binning gives the same image without any host read-back
    st = isect_tiles_static(ins[0], radii.to(dev), dep.to(dev), ts, tw, th, capacity=flat.numel() + 1000)
    assert int(st["n_isects"].item()) == flat.numel() and int(st["overflow"].item()) == 0
    assert torch.equal(st["flatten_ids"][:flat.numel()].cpu(), flat)
    c2, a2 = rasterize_to_pixels(*ins, W, H, ts, st["isect_offsets"], st["flatten_ids"], n_isects=st["n_isects"])
    assert torch.equal(c2, ref_c) and torch.equal(a2, ref_a)
    # overflow is reported, never written past capacity
    st2 = isect_tiles_static(ins[0], radii.to(dev), dep.to(dev), ts, tw, th, capacity=flat.numel() // 2)
    assert int(st2["overflow"].item()) == 1
    # tile masks: masked tiles render the background only
    masks = torch.ones(C, th, tw, dtype=torch.bool)
    masks[:, ::2, ::2] = False
    bg = torch.rand(C, 3)
    mc, ma = rasterize_to_pixels(*ins, W, H, ts, off.to(dev), flat.to(dev), backgrounds=bg.to(dev), masks=masks.to(dev))
    fc, fa = rasterize_to_pixels(*ins, W, H, ts, off.to(dev), flat.to(dev), backgrounds=bg.to(dev))
    pm = masks.repeat_interleave(ts, 1).repeat_interleave(ts, 2)[:, :H, :W].to(dev)
    assert torch.equal(mc[pm], fc[pm])
    assert torch.allclose(mc[~pm], bg.to(dev)[:, None, None, :].expand(C, H, W, 3)[~pm])
    assert ma[~pm].abs().max().item() == 0.0


def test_adam_matches_torch(dev):
    import ctypes
    from splat_one_amd import _lib
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 4), (1000,), (1000, 15, 3), (37,), (5, 1, 3)]
    lrs = [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 1.25e-4]
    b1, b2, eps = 0.9, 0.999, 1e-15
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_() for p in ps]
    opts = [torch.optim.Adam([r], lr=lr, betas=(b1, b2), eps=eps) for r, lr in zip(ref, lrs)]
    mine = [p.clone().to(dev) for p in ps]
    ms = [torch.zeros_like(p) for p in mine]
    vs = [torch.zeros_like(p) for p in mine]
    for step in range(1, 6):
        grads = [torch.randn(s, generator=g) * (10.0 ** (step - 3)) for s in shapes]
        gd = [x.clone().to(dev) for x in grads]
        for r, o, x in zip(ref, opts, grads):
            r.grad = x.clone()
            o.step()
        arr = (_lib.AdamGroup * len(mine))()
        for i, (p, x, m, v, lr) in enumerate(zip(mine, gd, ms, vs, lrs)):
            arr[i] = _lib.AdamGroup(p.data_ptr(), x.data_ptr(), m.data_ptr(), v.data_ptr(), 0, p.numel(), 1,
                                    lr / (1 - b1 ** step), math.sqrt(1 - b2 ** step))
        _lib.call("so_adam_step", len(mine), arr, b1, b2, eps, 1, _lib.stream())
        for x in gd:
            assert x.abs().max().item() == 0.0                 # zero_grad fused
        for p, r, o, m, v, lr in zip(mine, ref, opts, ms, vs, lrs):
            # parameters: within 2 ulp of the value plus 1e-6 of the update size (the update is ~lr)
            d = (p.cpu() - r.detach()).abs()
            assert (d <= 2.4e-7 * r.detach().abs() + 1e-6 * lr).all(), (step, d.max())
            st = o.state[r]
            assert torch.allclose(m.cpu(), st["exp_avg"], rtol=1e-6, atol=1e-30)
            assert torch.allclose(v.cpu(), st["exp_avg_sq"], rtol=1e-6, atol=1e-30)


@pytest.mark.parametrize("B,H,W,CH", [(1, 70, 90, 3), (2, 33, 47, 3), (1, 64, 64, 1), (1, 40, 40, 4)])
def test_photometric_loss(dev, B, H, W, CH):
    """Fused L1 + SSIM forward/backward vs the float64 torch restatement (autograd)."""
    from oracle import ssim_oracle as SO
    from splat_one_amd.losses import fused_ssim, photometric_loss
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.rand(B, H, W, CH, generator=g)
    y = (x + 0.2 * torch.randn(B, H, W, CH, generator=g)).clamp(0, 1)
    if CH == 3:
        xh = x.to(dev).requires_grad_()
        loss_h, l1_h, ss_h = photometric_loss(xh, y.to(dev), 0.2)
        (loss_h * 3.0).backward()
        xo = x.double().requires_grad_()
        loss_o, l1_o, ss_o = SO.photometric_loss(xo, y, 0.2)
        (loss_o * 3.0).backward()
        assert abs(loss_h.item() - loss_o.item()) < 2e-6
        assert abs(l1_h.item() - l1_o.item()) < 2e-6 and abs(ss_h.item() - ss_o.item()) < 2e-6
        assert rel_err(xh.grad, xo.grad) < 1e-4
    for padding in ("same", "valid"):
        xh = x.permute(0, 3, 1, 2).contiguous().to(dev).requires_grad_()
        v_h = fused_ssim(xh, y.permute(0, 3, 1, 2).to(dev), padding=padding)
        v_h.backward()
        xo = x.permute(0, 3, 1, 2).double().requires_grad_()
        v_o = SO.fused_ssim(xo, y.permute(0, 3, 1, 2), padding=padding)
        v_o.backward()
        assert abs(v_h.item() - v_o.item()) < 2e-6, padding
        assert rel_err(xh.grad, xo.grad) < 1e-4, padding


def test_wave_reduction_primitives(dev):
    """Transposing butterfly (8 sums in 26 instructions, slot s in lane with slot_of_lane==s) and the
    two all-lane sums, against a float64 sum of exact small integers (bitwise in fp32)."""
    from splat_one_amd import _lib
    g = torch.Generator().manual_seed(0)
    n = 37
    x = torch.randint(-50, 50, (n * 64, 9), generator=g).float()
    out = torch.zeros(n, 10, device=dev)
    xd = x.to(dev)
    _lib.call("so_debug_wave_reduce", n, _lib.ptr(xd), _lib.ptr(out), _lib.stream())
    ref = x.reshape(n, 64, 9).double().sum(1)
    assert torch.equal(out[:, :8].cpu().double(), ref[:, :8])
    assert torch.equal(out[:, 8].cpu().double(), ref[:, 8]) and torch.equal(out[:, 9].cpu().double(), ref[:, 8])


def test_wave_reduce9_network(dev):
    """The rasteriser backward's nine 64-lane sums as one transposing network (bank-masked DPP, permlane swaps, row_bcast):
    every slot against a float64 sum of exact small integers (bitwise in fp32), per-lane patterns that would expose a
    wrong partner lane (each lane a distinct power-of-two-free integer)."""
    from splat_one_amd import _lib
    g = torch.Generator().manual_seed(1)
    n = 41
    x = torch.randint(-50, 50, (n * 64, 9), generator=g).float()
    # first wave: column k of lane l = (k + 1) * (l + 1): a lane left out or counted twice changes the sum
    lanes = torch.arange(1, 65, dtype=torch.float32)
    x[:64] = lanes[:, None] * torch.arange(1, 10, dtype=torch.float32)[None]
    out = torch.zeros(n, 9, device=dev)
    xd = x.to(dev)
    _lib.call("so_debug_wave_reduce9", n, _lib.ptr(xd), _lib.ptr(out), _lib.stream())
    ref = x.reshape(n, 64, 9).double().sum(1)
    assert torch.equal(out.cpu().double(), ref), (out.cpu()[:2], ref[:2])


def test_quadrant_cull_never_misses(dev):
    """The rasterisers' per-(wave, candidate) culling -- the box of the alpha >= 1/255 region, then the exact test (minimum of
    sigma over the quadrant's pixel-centre rectangle on the ONE vertical and ONE horizontal edge the centre can see) --
    against a float64 minimisation over all four edges plus a dense sample of the rectangle: neither test may reject a
    rectangle in which alpha reaches 1/255 (a miss would silently drop a contribution), and the exact test must not
    accept rectangles that are clearly outside (it would only cost passes, but then it is not doing its job)."""
    import numpy as np
    from splat_one_amd import _lib
    rng = np.random.default_rng(5)
    n = 200_000
    # conics of random ellipses (radii 0.3 .. 40 px, any orientation), centres around an 8x8-pixel rectangle
    r1, r2 = np.exp(rng.uniform(np.log(0.3), np.log(40.0), (2, n)))
    th = rng.uniform(0, np.pi, n)
    c, s_ = np.cos(th), np.sin(th)
    ia, ib = 1.0 / r1 ** 2, 1.0 / r2 ** 2
    ca, cb, cc = c * c * ia + s_ * s_ * ib, c * s_ * (ia - ib), s_ * s_ * ia + c * c * ib
    op = np.where(rng.random(n) < 0.1, rng.uniform(0.0, 0.01, n), rng.uniform(0.004, 1.0, n))
    x0 = rng.integers(0, 200, n) + 0.5
    y0 = rng.integers(0, 200, n) + 0.5
    x1, y1 = x0 + 7.0, y0 + 7.0
    reach = np.maximum(r1, r2) * 4 + 12
    mx = x0 + 3.5 + rng.uniform(-1, 1, n) * reach
    my = y0 + 3.5 + rng.uniform(-1, 1, n) * reach
    rows = np.stack([mx, my, op, ca, cb, cc, x0, x1, y0, y1], 1).astype(np.float32)
    inp = torch.from_numpy(rows).to(dev)
    out = torch.empty(n, 2, device=dev)
    _lib.call("so_debug_cull", n, _lib.ptr(inp), _lib.ptr(out), _lib.stream())
    box_hit, exact_hit = (out[:, 0] > 0).cpu().numpy(), (out[:, 1] > 0).cpu().numpy()
    # float64 truth on the float32 inputs: minimum of sigma over the rectangle
    mx, my, op, ca, cb, cc, x0, x1, y0, y1 = rows.astype(np.float64).T
    ax0, ax1, ay0, ay1 = x0 - mx, x1 - mx, y0 - my, y1 - my

    def sig(x, y):
        return 0.5 * (ca * x * x + cc * y * y) + cb * x * y
    best = np.full(n, np.inf)
    for xe in (ax0, ax1):
        best = np.minimum(best, sig(xe, np.clip(-cb * xe / cc, ay0, ay1)))
    for ye in (ay0, ay1):
        best = np.minimum(best, sig(np.clip(-cb * ye / ca, ax0, ax1), ye))
    best = np.where((ax0 <= 0) & (ax1 >= 0) & (ay0 <= 0) & (ay1 >= 0), 0.0, best)
    for fx in np.linspace(0, 1, 8):          # the pixel centres themselves: what the rasteriser evaluates
        for fy in np.linspace(0, 1, 8):
            assert (sig(ax0 + 7 * fx, ay0 + 7 * fy) >= best * (1 - 1e-12) - 1e-12).all()
    with np.errstate(divide="ignore"):
        tau = np.log(255.0 * op)             # alpha >= 1/255  <=>  sigma <= tau
    reaches = (op * 255.0 >= 1.0) & (best <= tau)
    assert reaches.sum() > 20_000 and (~reaches).sum() > 20_000
    assert not (reaches & ~box_hit).any(), "the box test rejected a rectangle that is hit"
    assert not (reaches & ~exact_hit).any(), "the exact test rejected a rectangle that is hit"
    clearly_out = (op * 255.0 < 0.99) | (best > tau * 1.01 + 0.05)
    assert not (clearly_out & exact_hit).any(), "the exact test accepted a rectangle that is clearly missed"
    assert (box_hit & ~exact_hit).sum() > 1000      # it does remove what the box lets through


def test_camera_inverse(dev):
    from splat_one_amd import _lib
    from splat_one_amd.scene import ring_cameras
    c2w = ring_cameras(8, 9.0, 0.7)
    c2w[:, :3, :3] *= torch.linspace(0.5, 2.0, 8)[:, None, None]      # non-rigid (scaled) poses too
    out = torch.empty(8, 4, 4, device=dev)
    src = c2w.to(dev).contiguous()
    _lib.call("so_camera_inverse", 8, _lib.ptr(src), _lib.ptr(out), _lib.stream())
    ref = torch.linalg.inv(c2w.double())
    assert torch.allclose(out.cpu().double(), ref, rtol=1e-6, atol=1e-6)


def test_step_inputs_one_launch(dev):
    """so_step_inputs: camera inverse (vs float64 inverse), intrinsics copy, target-image slot, counter zeroing
    and the Adam schedule (vs the closed form torch.optim.Adam + ExponentialLR use)."""
    import ctypes
    from splat_one_amd import _lib
    C, nz, ng = 5, 3001, 3
    g = torch.Generator().manual_seed(3)
    c2w = torch.eye(4).repeat(C, 1, 1)
    c2w[:, :3, :3] = torch.linalg.qr(torch.randn(C, 3, 3, generator=g))[0]
    c2w[:, :3, 3] = torch.randn(C, 3, generator=g) * 4
    c2w[0, :3, :3] *= 1.7                                        # not rigid: the inverse is the general one
    Ks = torch.rand(C, 3, 3, generator=g)
    c2w_d, Ks_d = c2w.to(dev), Ks.to(dev)
    vm, Kd = torch.empty(C, 4, 4, device=dev), torch.empty(C, 3, 3, device=dev)
    img = torch.rand(2, 3, device=dev)
    slot = torch.zeros(1, dtype=torch.int64, device=dev)
    counters = torch.full((nz + 7,), 77, dtype=torch.int32, device=dev)
    step_dev = torch.zeros(2 + 4 * _lib.SO_ADAM_MAX_GROUPS, dtype=torch.int32, device=dev)
    step_dev[0] = 41
    lr0 = (ctypes.c_float * ng)(1.6e-4, 5e-3, 1e-3)
    gam = (ctypes.c_float * ng)(0.99985, 1.0, 1.0)
    b1, b2 = 0.9, 0.999
    _lib.call("so_step_inputs", C, _lib.ptr(c2w_d), _lib.ptr(Ks_d), _lib.ptr(vm), _lib.ptr(Kd), _lib.ptr(img), _lib.ptr(slot),
              _lib.ptr(counters), nz, ng, lr0, gam, b1, b2, _lib.ptr(step_dev), 0, 0, 0, 0, 0, 0, 0, 0, _lib.stream())
    torch.cuda.synchronize()
    want = torch.linalg.inv(c2w.double())
    assert (vm.cpu().double() - want).abs().max().item() < 1e-6
    assert torch.equal(Kd.cpu(), Ks)
    assert slot.item() == img.data_ptr()
    assert (counters[:nz] == 0).all() and (counters[nz:] == 77).all()
    assert step_dev[0].item() == 42                               # schedule evaluated for step 41, counter advanced
    hyper = step_dev[2:2 + 2 * ng].view(torch.float32).cpu().reshape(ng, 2)
    for i in range(ng):
        lr = lr0[i] * gam[i] ** 41
        assert abs(hyper[i, 0].item() - lr / (1 - b1 ** 42)) <= 1e-6 * lr / (1 - b1 ** 42)
        assert abs(hyper[i, 1].item() - math.sqrt(1 - b2 ** 42)) < 1e-6
    # status of the previous iteration: published to host-mapped memory before the pair is zeroed
    status = torch.zeros(4, dtype=torch.int32).pin_memory()
    counters.fill_(9)
    counters[100], counters[101] = 123456, 1
    _lib.call("so_step_inputs", 0, 0, 0, 0, 0, 0, 0, _lib.ptr(counters), nz, 0, None, None, 0.0, 0.0, 0,
              status.data_ptr(), 100, 77, 0, 0, 0, 0, 0, _lib.stream())
    torch.cuda.synchronize()
    assert status[:3].tolist() == [123456, 1, 77] and (counters[:nz] == 0).all()
    # per-tile list lengths (the first n_lists counters): maximum and sum gathered while they are zeroed, published by the
    # NEXT call -- whatever the counters hold then -- and the pair is free again for the call after that
    status = torch.zeros(5, dtype=torch.int32).pin_memory()
    stat = torch.zeros(4, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(3)
    seen = []
    for seq in (5, 6, 7, 8):
        lens = torch.randint(0, 3000, (90,), generator=g, dtype=torch.int32)
        counters.fill_(4)
        counters[:90] = lens.to(dev)
        _lib.call("so_step_inputs", 0, 0, 0, 0, 0, 0, 0, _lib.ptr(counters), nz, 0, None, None, 0.0, 0.0, 0,
                  status.data_ptr(), 100, seq, 90, _lib.ptr(stat), 0, 0, 0, _lib.stream())
        torch.cuda.synchronize()
        assert int(status[2]) == seq and (counters[:nz] == 0).all()
        if seen:
            assert status[3:5].tolist() == [int(seen[-1].max()), int(seen[-1].sum())], seq
        seen.append(lens)
    # parts are optional: zero only
    counters.fill_(5)
    _lib.call("so_step_inputs", 0, 0, 0, 0, 0, 0, 0, _lib.ptr(counters), 10, 0, None, None, 0.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, _lib.stream())
    torch.cuda.synchronize()
    assert (counters[:10] == 0).all() and (counters[10:] == 5).all() and step_dev[0].item() == 42
    # order_src: a kept workgroup -> tile table copied into the step's table by the same launch
    src = torch.randperm(5000, device=dev).to(torch.int32)
    dst = torch.zeros(5000, dtype=torch.int32, device=dev)
    _lib.call("so_step_inputs", 0, 0, 0, 0, 0, 0, 0, _lib.ptr(counters), nz, 0, None, None, 0.0, 0.0, 0, 0, 0, 0, 0, 0,
              _lib.ptr(src), _lib.ptr(dst), 5000, _lib.stream())
    torch.cuda.synchronize()
    assert torch.equal(src, dst)


def test_rec_unpack_and_cam_stride(dev):
    """so_rec_unpack reads (centre, depth, radius) out of the 64-byte records; so_preprocess_fwd with a camera
    stride larger than N writes the same rows as the dense call, and a NULL histogram skips the binning pass."""
    from splat_one_amd import _lib
    from splat_one_amd.scene import make_scene
    W, H, N, C, cap = 160, 120, 500, 2, 640
    splats, c2w, Ks = make_scene(N, W, H, "ref", n_views=C)
    s = {k: v.to(dev).contiguous() for k, v in splats.items()}
    vm = torch.linalg.inv(c2w).to(dev).contiguous()
    Kd = Ks.to(dev).contiguous()
    K = 1 + s["shN"].shape[1]
    p = _lib.ptr

    def run(stride, hist):
        n_rows = C * stride
        o = dict(radii=torch.zeros(n_rows, dtype=torch.int32, device=dev), means2d=torch.zeros(n_rows, 2, device=dev),
                 depths=torch.zeros(n_rows, device=dev), conics=torch.zeros(n_rows, 3, device=dev),
                 opac=torch.zeros(n_rows, device=dev), colors=torch.zeros(n_rows, 3, device=dev),
                 tpg=torch.zeros(n_rows, dtype=torch.int32, device=dev), rec=torch.zeros(n_rows, 16, device=dev),
                 hist=torch.zeros(C * 8 * 10, dtype=torch.int32, device=dev))
        _lib.call("so_preprocess_fwd", C, N, K, 3, p(s["means"]), p(s["scales"]), p(s["quats"]), p(s["opacities"]), p(s["sh0"]),
                  p(s["shN"]), p(vm), p(Kd), W, H, 0.3, 0.01, 1e8, 0.0, 0, 0, 16, p(o["radii"]), p(o["means2d"]), p(o["depths"]),
                  p(o["conics"]), p(o["opac"]), p(o["colors"]), p(o["tpg"]), p(o["hist"]) if hist else 0, p(o["rec"]), 0,
                  stride, 0, 0, 0, 0, 0, _lib.stream())
        return o
    dense, strided = run(N, True), run(cap, False)
    assert dense["hist"].sum().item() == dense["tpg"].sum().item() > 0 and strided["hist"].sum().item() == 0
    for k in ("radii", "means2d", "depths", "conics", "opac", "colors", "tpg", "rec"):
        a = dense[k].reshape(C, N, -1)
        b = strided[k].reshape(C, cap, -1)
        assert torch.equal(a, b[:, :N]), k
        assert (b[:, N:] == 0).all(), k                      # padding rows untouched
    rec = strided["rec"]
    m2, rad, dep = torch.empty(C * cap, 2, device=dev), torch.empty(C * cap, dtype=torch.int32, device=dev), torch.empty(C * cap, device=dev)
    vrec = torch.ones(C * cap, 16, device=dev)
    _lib.call("so_rec_unpack", C * cap, p(rec), p(m2), p(rad), p(dep), p(vrec), _lib.stream())
    assert torch.equal(m2, strided["means2d"]) and torch.equal(rad, strided["radii"]) and torch.equal(dep, strided["depths"])
    assert (vrec == 0).all()


@pytest.mark.parametrize("cull", [0, 1])
def test_slotted_binning_and_exact_tile_cull_c_abi(dev, cull):
    """so_preprocess_fwd(tile_slots[, tile_cull]) -> so_isect_scan(both histogram halves) -> so_isect_fill(tile_slots
    [, cull_rec]) through the C ABI.  Without culling every tile's list equals, as a sorted sequence, the list of the
    plain path (so_isect_count / so_isect_fill).  With culling every list is a subsequence of it, and every pair that
    was dropped has alpha < 1/255 at ALL 256 pixel centres of its tile (float64 brute force): the cull is exact."""
    from splat_one_amd import _lib
    from splat_one_amd.scene import make_scene
    W, H, N, C = 320, 208, 3000, 2
    splats, c2w, Ks = make_scene(N, W, H, "ref", n_views=C)
    g = torch.Generator().manual_seed(11)
    splats["scales"] = splats["scales"] + torch.randn(N, 3, generator=g) * 0.7       # needles, blobs, > 12-tile rectangles
    splats["scales"][:200] -= 2.0                                                     # sub-pixel ones
    splats["opacities"] = torch.randn(N, generator=g) * 3.0 - 2.0                     # many near / below 1/255
    s = {k: v.to(dev).contiguous() for k, v in splats.items()}
    vm, Kd = torch.linalg.inv(c2w).to(dev).contiguous(), Ks.to(dev).contiguous()
    K, ts = 1 + s["shN"].shape[1], 16
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    M = C * tw * th
    p, st = _lib.ptr, _lib.stream()
    z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=dev)
    o = dict(radii=z(C * N, dt=torch.int32), means2d=z(C * N, 2), depths=z(C * N), conics=z(C * N, 3), opac=z(C * N),
             colors=z(C * N, 3), tpg=z(C * N, dt=torch.int32), rec=z(C * N, 16), slots=z(C * N, _lib.SO_TILE_SLOTS, dt=torch.int32))
    counters = z(2 * M + 4, dt=torch.int32)                   # small histogram | large-rectangle histogram | long-list length, n_isects, overflow
    offsets, n_is, ovf = z(M, dt=torch.int32), counters[2 * M + 1:], counters[2 * M + 2:]
    _lib.call("so_preprocess_fwd", C, N, K, 3, p(s["means"]), p(s["scales"]), p(s["quats"]), p(s["opacities"]), p(s["sh0"]), p(s["shN"]),
              p(vm), p(Kd), W, H, 0.3, 0.01, 1e8, 0.0, 0, 0, ts, p(o["radii"]), p(o["means2d"]), p(o["depths"]), p(o["conics"]),
              p(o["opac"]), p(o["colors"]), p(o["tpg"]), p(counters), p(o["rec"]), 0, 0, p(o["slots"]), cull, 0, 0, 0, st)
    assert counters[M:2 * M].sum().item() > 0 and counters[:M].sum().item() > 0           # both kinds of rectangle occur
    _lib.call("so_isect_scan", C, tw, th, p(counters), p(counters[M:]), p(offsets), p(n_is), st)
    n = int(n_is[0].item())
    cap = n + 16
    keys, flat = z(cap, dt=torch.int64), torch.full((cap,), -1, dtype=torch.int32, device=dev)
    _lib.call("so_isect_fill", C, N, p(o["means2d"]), p(o["radii"]), p(o["depths"]), ts, tw, th, p(offsets), p(n_is),
              p(counters[M:]), cap, p(keys), p(flat), 0, p(ovf), p(o["slots"]), p(o["rec"]) if cull else 0, st)
    assert int(ovf[0].item()) == 0 and (flat[:n] >= 0).all() and (flat[:n] < C * N).all()
    # the plain path on the same projected Gaussians
    from splat_one_amd.ops import isect_offset_encode, isect_tiles
    _, ids_ref, flat_ref = isect_tiles(o["means2d"].view(C, N, 2), o["radii"].view(C, N), o["depths"].view(C, N), ts, tw, th)
    offs_ref = isect_offset_encode(ids_ref, C, tw, th).reshape(-1).cpu().tolist() + [flat_ref.numel()]
    offs = offsets.cpu().tolist() + [n]
    flat, flat_ref = flat.cpu(), flat_ref.cpu()
    if not cull:
        assert n == flat_ref.numel() and offs == offs_ref and torch.equal(flat[:n], flat_ref)
        return
    assert 0.2 * flat_ref.numel() < n < 0.8 * flat_ref.numel(), (n, flat_ref.numel())
    m2, con, op = o["means2d"].cpu().double(), o["conics"].cpu().double(), o["opac"].cpu().double()
    py, px = torch.meshgrid(torch.arange(ts, dtype=torch.float64) + 0.5, torch.arange(ts, dtype=torch.float64) + 0.5, indexing="ij")
    n_dropped, worst = 0, 0.0
    for t in range(M):
        full = flat_ref[offs_ref[t]:offs_ref[t + 1]].tolist()
        kept = flat[offs[t]:offs[t + 1]].tolist()
        it = iter(full)
        assert all(any(k == f for f in it) for k in kept), t            # an order-preserving subsequence
        dropped = sorted(set(full) - set(kept))
        if not dropped:
            continue
        tile = t % (tw * th)
        x0, y0 = (tile % tw) * ts, (tile // tw) * ts
        d = torch.tensor(dropped)
        dx = (x0 + px)[None] - m2[d, 0, None, None]
        dy = (y0 + py)[None] - m2[d, 1, None, None]
        sigma = 0.5 * (con[d, 0, None, None] * dx * dx + con[d, 2, None, None] * dy * dy) + con[d, 1, None, None] * dx * dy
        alpha = op[d, None, None] * torch.exp(-sigma)
        alpha = torch.where(sigma < 0, torch.zeros_like(alpha), alpha)
        worst = max(worst, alpha.max().item())
        n_dropped += len(dropped)
    assert n_dropped == flat_ref.numel() - n
    assert worst < 1.0 / 255.0, worst


def test_camera_inverse_matches_torch_and_is_differentiable(dev):
    """ops.camera_inverse == torch.linalg.inv on rigid and on general 4x4 matrices, with the analytic gradient
    (-A^-T v A^-T) that pose optimisation relies on -- and no device synchronisation."""
    from splat_one_amd.ops import camera_inverse
    from splat_one_amd.scene import ring_cameras
    g = torch.Generator().manual_seed(3)
    A = torch.cat([ring_cameras(5), torch.eye(4)[None] + 0.3 * torch.randn(3, 4, 4, generator=g)]).to(dev).requires_grad_(True)
    B = A.detach().clone().requires_grad_(True)
    w = torch.randn(8, 4, 4, generator=g).to(dev)
    inv_h, inv_t = camera_inverse(A), torch.linalg.inv(B.double()).float()
    assert (inv_h - inv_t).abs().max().item() < 1e-5
    (inv_h * w).sum().backward()
    (torch.linalg.inv(B) * w).sum().backward()
    assert rel_err(A.grad, B.grad) < 1e-4


def test_tile_cull_count_and_fill_passes_agree(dev, monkeypatch):
    """Exact tile culling is evaluated twice -- by the histogram pass that sizes the lists and by the fill pass that
    writes them -- in two different kernels, and the two must agree bit for bit: with exact-size lists a pair that is
    counted but not filled is an unwritten slot.  (Round 2: the compiler had fused the test's multiplies and adds into
    FMAs in one kernel only; one disagreement per ~2e8 tests.  so_common.hpp now pins the rounding.)  ~4e8 borderline-rich
    tests: random anisotropic conics, opacities down to the 1/255 threshold, centres on a 1080p image."""
    from splat_one_amd.ops import isect_tiles, rec_pack_unpack_roundtrip
    monkeypatch.setenv("SPLAT_ONE_AMD_CHECK_LISTS", "1")       # isect_tiles raises if any list slot stays unwritten
    W, H, ts = 1920, 1080, 16
    tw, th = W // ts, (H + ts - 1) // ts
    N = 2_000_000
    g = torch.Generator(device=dev).manual_seed(99)
    n_pairs = 0
    for rnd in range(12):
        means2d = torch.rand(1, N, 2, generator=g, device=dev) * torch.tensor([W, H], device=dev)
        s1 = torch.rand(1, N, generator=g, device=dev) * 12 + 0.6
        s2 = s1 * (torch.rand(1, N, generator=g, device=dev) * 0.9 + 0.1)
        th_ = torch.rand(1, N, generator=g, device=dev) * math.pi
        c, s = torch.cos(th_), torch.sin(th_)
        i1, i2 = 1.0 / (s1 * s1), 1.0 / (s2 * s2)
        conics = torch.stack([c * c * i1 + s * s * i2, c * s * (i1 - i2), s * s * i1 + c * c * i2], -1).contiguous()
        radii = torch.ceil(3.0 * s1).to(torch.int32)
        opac = (torch.rand(1, N, generator=g, device=dev) ** 3) * 0.99 + 0.004
        depths = torch.rand(1, N, generator=g, device=dev) + 1.0
        tpg, ids, flat = isect_tiles(means2d, radii, depths, ts, tw, th, conics=conics, opacities=opac)
        _, ids0, _ = isect_tiles(means2d, radii, depths, ts, tw, th)
        assert 0.2 * ids0.numel() < ids.numel() < ids0.numel()
        n_pairs += ids0.numel()
    assert n_pairs > 3e8
    assert rec_pack_unpack_roundtrip(dev)


@pytest.mark.parametrize("degree,C", [(3, 1), (2, 3), (0, 2)])
def test_sh_view_colors_equals_the_operator_chain(dev, degree, C):
    """`sh_view_colors` (one launch each way) against what `rasterization` otherwise composes from six operators:
    clamp_min(spherical_harmonics(deg, means - campos, coeffs, masks = radii > 0) + 0.5, 0) -- values and gradients."""
    from splat_one_amd.ops import sh_view_colors, spherical_harmonics
    N, K = 5000, 16
    g = torch.Generator().manual_seed(13)
    means0 = torch.randn(N, 3, generator=g)
    coeffs0 = torch.randn(N, K, 3, generator=g) * 2.0
    campos = (torch.randn(C, 3, generator=g) * 3).to(dev)
    radii = (torch.rand(C, N, generator=g) < 0.8).to(torch.int32).to(dev) * 7
    w = torch.rand(C, N, 3, generator=g).to(dev)
    outs = []
    for fused in (False, True):
        means = means0.clone().to(dev).requires_grad_(True)
        coeffs = coeffs0.clone().to(dev).requires_grad_(True)
        if fused:
            col = sh_view_colors(degree, means, campos, coeffs, radii)
        else:
            dirs = means[None] - campos[:, None]
            col = torch.clamp_min(spherical_harmonics(degree, dirs, coeffs[None].expand(C, -1, -1, -1), masks=radii > 0) + 0.5, 0.0)
        (col * w).sum().backward()
        outs.append((col.detach(), means.grad.clone(), coeffs.grad.clone()))
    (c0, gm0, gc0), (c1, gm1, gc1) = outs
    assert (c0 - c1).abs().max().item() <= 2e-6
    assert float((c1 == 0).float().mean()) > 0.01                        # the clamp is exercised
    assert rel_err(gm1, gm0) < 1e-5 and rel_err(gc1, gc0) < 1e-6
    # and against the float64 oracle
    col_o = torch.clamp_min(O.spherical_harmonics(degree, (means0[None].double() - campos.cpu().double()[:, None]),
                                                  coeffs0[None].double().expand(C, -1, -1, -1), masks=(radii > 0).cpu()) + 0.5, 0.0)
    assert (c1.cpu().double() - col_o).abs().max().item() < 1e-5


def test_strategy_update_state_kernel(dev):
    """`so_strategy_update_state` (DefaultStrategy._update_state on the dense layout, one launch) against the loop oracle."""
    from oracle import strategy_oracle as SO
    from splat_one_amd.strategy import DefaultStrategy
    C, N, W, H = 3, 4000, 640, 360
    g = torch.Generator().manual_seed(2)
    grads = torch.randn(C, N, 2, generator=g) * 1e-3
    radii = (torch.rand(C, N, generator=g) < 0.6).to(torch.int32) * torch.randint(1, 40, (C, N), generator=g, dtype=torch.int32)
    s = DefaultStrategy(refine_scale2d_stop_iter=4000)
    state = {"grad2d": torch.rand(N, generator=g).to(dev), "count": torch.randint(0, 5, (N,), generator=g).float().to(dev),
             "radii": torch.rand(N, generator=g).to(dev) * 0.01, "scene_scale": 1.0}
    g2_0, cn_0, r_0 = (state[k].cpu().double().clone() for k in ("grad2d", "count", "radii"))
    m2 = torch.zeros(C, N, 2, device=dev, requires_grad=True)
    m2.grad = grads.to(dev).contiguous()
    info = {"width": W, "height": H, "n_cameras": C, "radii": radii.to(dev), "means2d": m2}
    s._update_state({"means": torch.zeros(N, 3)}, state, info)
    g2, cn = SO.update_state(g2_0, cn_0, grads.double(), radii, W, H, C)
    assert (state["grad2d"].cpu().double() - g2).abs().max().item() < 1e-5
    assert torch.equal(state["count"].cpu().double(), cn)
    r_want = torch.maximum(r_0, (radii.double() / max(W, H)).max(dim=0).values * (radii > 0).any(dim=0))
    assert (state["radii"].cpu().double() - r_want).abs().max().item() < 1e-7


@pytest.mark.parametrize("seed", list(range(16)))
def test_isect_random_inputs_bit_exact(dev, seed):
    """`isect_tiles` on random centres / radii / depths -- image sizes that are and are not multiples of the tile, both
    tile sizes, several cameras, periodic images with footprints up to wider than the image -- bit for bit against the
    oracle: counts, sorted keys, ids, offsets, and the unsorted emission order."""
    import random
    from splat_one_amd.ops import isect_tiles
    rnd = random.Random(70 + seed)
    ts = rnd.choice([16, 16, 8])
    periodic = rnd.random() < 0.5
    tw, th = rnd.randint(2, 14), rnd.randint(1, 9)
    W, H = (tw * ts, th * ts - rnd.randint(0, ts - 1)) if periodic else (tw * ts - rnd.randint(0, ts - 1), th * ts - rnd.randint(0, ts - 1))
    C, N = rnd.choice([1, 2, 3]), rnd.choice([1, 17, 300, 2500])
    g = torch.Generator().manual_seed(800 + seed)
    m2 = torch.rand(C, N, 2, generator=g) * torch.tensor([W * 1.2, H * 1.2]) - torch.tensor([W * 0.1, H * 0.1])
    if periodic:
        m2[..., 0] = m2[..., 0].remainder(W)          # a panorama's centres lie inside the image
    big = rnd.choice([6, 40, int(0.8 * W)])
    radii = (torch.rand(C, N, generator=g) ** 2 * big).to(torch.int32) * (torch.rand(C, N, generator=g) < 0.85)
    radii = radii.to(torch.int32)
    dep = torch.rand(C, N, generator=g) * 5 + 0.1
    dep[:, ::7] = dep[:, :1].clone()                 # depth ties: the order falls back on the id
    tpg_o, ids_o, flat_o = O.isect_tiles(m2, radii, dep, ts, tw, th, periodic=periodic)
    off_o = O.isect_offset_encode(ids_o, C, tw, th)
    tpg_h, ids_h, flat_h, off_h = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), ts, tw, th, return_offsets=True, periodic=periodic)
    assert torch.equal(tpg_h.cpu(), tpg_o) and torch.equal(ids_h.cpu(), ids_o) and torch.equal(flat_h.cpu(), flat_o)
    assert torch.equal(off_h.cpu(), off_o)
    _, ids_u_o, flat_u_o = O.isect_tiles(m2, radii, dep, ts, tw, th, sort=False, periodic=periodic)
    _, ids_u_h, flat_u_h = isect_tiles(m2.to(dev), radii.to(dev), dep.to(dev), ts, tw, th, sort=False, periodic=periodic)
    assert torch.equal(ids_u_h.cpu(), ids_u_o) and torch.equal(flat_u_h.cpu(), flat_u_o)


@pytest.mark.parametrize("seed", list(range(14)))
def test_photometric_loss_random_sizes(dev, seed):
    """L1 + SSIM forward / backward on random image sizes around the kernels' strip boundaries (256 floats of a row, 36
    rows), down to images smaller than the 11x11 window, batch and channel counts, both paddings, several lambdas."""
    import random
    from splat_one_amd.losses import fused_ssim, photometric_loss
    from oracle import ssim_oracle as SO
    rnd = random.Random(900 + seed)
    CH = rnd.choice([3, 3, 1, 4])
    B = rnd.choice([1, 1, 2, 3])
    W = rnd.choice([11, 12, 23, 85, 86, 255 // CH, 256 // CH + 1, 171, rnd.randint(11, 300)])
    H = rnd.choice([11, 13, 35, 36, 37, 46, 72, 73, rnd.randint(11, 150)])
    lam = rnd.choice([0.2, 0.0, 1.0, 0.5])
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, H, W, CH, generator=g)
    y = (x + 0.2 * torch.randn(B, H, W, CH, generator=g)).clamp(0, 1)
    if CH == 3:
        xh = x.to(dev).requires_grad_()
        loss_h, l1_h, ss_h = photometric_loss(xh, y.to(dev), lam)
        (loss_h * 2.0).backward()
        xo = x.double().requires_grad_()
        loss_o, l1_o, ss_o = SO.photometric_loss(xo, y, lam)
        (loss_o * 2.0).backward()
        assert abs(loss_h.item() - loss_o.item()) < 3e-6 and abs(l1_h.item() - l1_o.item()) < 3e-6 and abs(ss_h.item() - ss_o.item()) < 3e-6
        assert rel_err(xh.grad, xo.grad) < 1e-4, (B, H, W, lam)
    padding = rnd.choice(["same", "valid"])
    xh = x.permute(0, 3, 1, 2).contiguous().to(dev).requires_grad_()
    v_h = fused_ssim(xh, y.permute(0, 3, 1, 2).contiguous().to(dev), padding=padding)
    v_h.backward()
    xo = x.permute(0, 3, 1, 2).double().requires_grad_()
    v_o = SO.fused_ssim(xo, y.permute(0, 3, 1, 2), padding=padding)
    v_o.backward()
    assert abs(v_h.item() - v_o.item()) < 3e-6, (B, CH, H, W, padding)
    assert rel_err(xh.grad, xo.grad) < 1e-4, (B, CH, H, W, padding)


@pytest.mark.parametrize("seed", list(range(16)))
def test_fused_loss_kernel(dev, seed):
    """so_ssim_l1_fused (loss + gradient in one launch, derivative values through LDS) against the float64 oracle and
    against the forward/backward pair, on random sizes around its strip boundaries (512 - 10 CH floats of a row, any
    number of rows per workgroup), batches, channel counts, both paddings; every launch is repeated on the same work
    buffer (the loss scalars come from the workgroup that draws the last ticket, which must leave the ticket at zero)."""
    import random
    from splat_one_amd import _lib
    from splat_one_amd.losses import photometric_loss, photometric_loss_and_grad
    from oracle import ssim_oracle as SO
    rnd = random.Random(4100 + seed)
    CH = rnd.choice([3, 3, 3, 1, 4])
    B = rnd.choice([1, 1, 2, 3])
    out_t = 512 - 10 * CH           # floats of a row one workgroup emits
    W = rnd.choice([11, 12, 23, out_t // CH, out_t // CH + 1, 2 * out_t // CH, 2 * out_t // CH + 1, rnd.randint(11, 400)])
    H = rnd.choice([11, 13, 24, 25, 47, 48, 49, rnd.randint(11, 200)])
    rows = rnd.choice([0, 0, 1, 5, 11, 24, 57, H])
    lam = rnd.choice([0.2, 0.0, 1.0, 0.5])
    valid = rnd.choice([True, True, False])
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, H, W, CH, generator=g)
    y = (x + 0.2 * torch.randn(B, H, W, CH, generator=g)).clamp(0, 1)
    xd, yd = x.to(dev), y.to(dev)
    n_l1, n_ss = float(B * H * W * CH), float(B * CH * ((H - 10) * (W - 10) if valid else H * W))
    work = torch.zeros(6, device=dev)
    grad = torch.full_like(xd, float("nan"))
    up = torch.tensor([1.7], device=dev)
    outs = []
    for rep in range(2):
        work[:2].zero_()
        _lib.call("so_ssim_l1_fused", B, H, W, CH, _lib.ptr(xd), _lib.ptr(yd), 1 if valid else 0, (1.0 - lam) / n_l1, -lam / n_ss,
                  _lib.ptr(up), _lib.ptr(work), _lib.ptr(grad), _lib.ptr(work[2:]), _lib.ptr(work[5:]), lam, rows, _lib.stream())
        torch.cuda.synchronize()
        assert work[5:].view(torch.int32).item() == 0, "ticket not returned to zero"
        outs.append((work[2:5].cpu().clone(), grad.cpu().clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and (outs[0][0] - outs[1][0]).abs().max().item() < 1e-6
    # the float64 oracle: a * mean|x-y| + b * mean SSIM + c and its gradient, times the upstream 1.7
    xo = x.double().requires_grad_()
    l1_o = (xo - y.double()).abs().mean()
    ss_o = SO.fused_ssim(xo.permute(0, 3, 1, 2), y.permute(0, 3, 1, 2), padding="valid" if valid else "same")
    loss_o = (1.0 - lam) * l1_o + lam * (1.0 - ss_o)
    (loss_o * 1.7).backward()
    loss_h, grad_h = outs[0]
    assert abs(loss_h[0].item() - loss_o.item()) < 3e-6 and abs(loss_h[1].item() - l1_o.item()) < 3e-6, (B, CH, H, W, rows)
    assert abs(loss_h[2].item() - (1.0 - ss_o.item())) < 3e-6, (B, CH, H, W, rows)
    assert torch.isfinite(grad_h).all()
    assert rel_err(grad_h, xo.grad) < 1e-4, (B, CH, H, W, rows, lam, valid)
    if CH == 3 and valid:       # the Python front ends of both formulations agree to rounding
        xh = xd.clone().requires_grad_()
        loss_p, l1_p, ss_p = photometric_loss(xh, yd, lam)
        loss_p.backward()
        loss_f, l1_f, ss_f, grad_f = photometric_loss_and_grad(xd, yd, lam, rows=rows)
        assert abs(loss_p.item() - loss_f.item()) < 1e-6 and abs(l1_p.item() - l1_f.item()) < 1e-6 and abs(ss_p.item() - ss_f.item()) < 1e-6
        assert rel_err(grad_f, xh.grad) < 1e-5
