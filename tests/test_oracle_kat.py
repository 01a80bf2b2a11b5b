"""Known-answer tests that pin the oracle itself.  The reference ships no
test or golden vector for this path ("parity unpinned"), so these analytic cases are the anchor."""
import math

import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as O

dt = torch.float64
pytestmark = pytest.mark.filterwarnings("ignore::DeprecationWarning")


def _cam(W, H, f):
    K = torch.tensor([[[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]]], dtype=dt)
    return torch.eye(4, dtype=dt)[None], K


def _single(mean, sigma, opac, rgb, W=32, H=32, f=40.0, **kw):
    viewmats, Ks = _cam(W, H, f)
    means = torch.tensor([mean], dtype=dt)
    quats = torch.tensor([[1.0, 0, 0, 0]], dtype=dt)
    scales = torch.full((1, 3), sigma, dtype=dt)
    return O.rasterization(means, quats, scales, torch.tensor([opac], dtype=dt), torch.tensor([rgb], dtype=dt),
                           viewmats, Ks, W, H, **kw)


def test_kat1_single_isotropic_gaussian_closed_form():
    """pixel = o * exp(-d^2 / (2 (sigma^2 f^2/z^2 + 0.3))) * c on the optical axis; alpha capped at 0.999."""
    W = H = 32
    f, z, sigma, o = 40.0, 5.0, 0.25, 0.8
    rc, ra, meta = _single([0.0, 0.0, z], sigma, o, [0.2, 0.5, 0.9], W, H, f)
    var = sigma ** 2 * f ** 2 / z ** 2 + 0.3
    assert meta["radii"][0, 0].item() == math.ceil(3 * math.sqrt(var))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt) + 0.5, torch.arange(W, dtype=dt) + 0.5, indexing="ij")
    d2 = (xs - W / 2) ** 2 + (ys - H / 2) ** 2
    alpha = o * torch.exp(-0.5 * d2 / var)
    # only tiles overlapped by the 3-sigma AABB receive the splat; inside them alpha >= 1/255 is required
    r = meta["radii"][0, 0].item()
    tile_ok = torch.zeros(H, W, dtype=torch.bool)
    for ty in range(2):
        for tx in range(2):
            if (W / 2 + r > tx * 16) and (W / 2 - r < (tx + 1) * 16) and (H / 2 + r > ty * 16) and (H / 2 - r < (ty + 1) * 16):
                tile_ok[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    alpha = torch.where((alpha >= 1 / 255) & tile_ok, alpha, torch.zeros_like(alpha))
    assert torch.allclose(ra[0, ..., 0], alpha, atol=1e-12)
    assert torch.allclose(rc[0, ..., 1], alpha * 0.5, atol=1e-12)
    # alpha cap
    rc2, ra2, _ = _single([0.0, 0.0, z], 2.0, 1.0, [1.0, 1.0, 1.0], W, H, f)
    assert abs(ra2[0, H // 2, W // 2, 0].item() - 0.999) < 1e-9


def test_kat2_depth_order():
    """Two coincident splats at different depths: front one first; swapping depths changes the image."""
    W = H = 16
    viewmats, Ks = _cam(W, H, 20.0)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 2, dtype=dt)
    scales = torch.full((2, 3), 0.5, dtype=dt)
    opac = torch.tensor([0.6, 0.7], dtype=dt)
    cols = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]], dtype=dt)
    outs = []
    for z0, z1 in ((3.0, 4.0), (4.0, 3.0)):
        means = torch.tensor([[0, 0, z0], [0, 0, z1]], dtype=dt)
        rc, ra, meta = O.rasterization(means, quats, scales, opac, cols, viewmats, Ks, W, H)
        outs.append((rc, ra, meta))
    (rc_a, ra_a, m_a), (rc_b, _, m_b) = outs
    p = (H // 2, W // 2)
    conic = m_a["conics"][0]
    # closed form at the centre pixel (offset 0.5,0.5 from the mean)
    def alpha(i, meta):
        c = meta["conics"][0, i]
        s = 0.5 * (c[0] * 0.25 + c[2] * 0.25) + c[1] * 0.25
        return float(min(0.999, opac[i] * math.exp(-s)))
    a0, a1 = alpha(0, m_a), alpha(1, m_a)
    assert abs(rc_a[0][p][0].item() - a0) < 1e-12                   # red in front
    assert abs(rc_a[0][p][1].item() - a1 * (1 - a0)) < 1e-12        # green attenuated by T
    assert abs(ra_a[0][p][0].item() - (1 - (1 - a0) * (1 - a1))) < 1e-12
    assert m_a["flatten_ids"].tolist()[0] == 0 and m_b["flatten_ids"].tolist()[0] == 1
    assert (rc_a - rc_b).abs().max() > 0.05


def test_kat3_culling_gives_zero_radii_and_grads():
    W = H = 32
    viewmats, Ks = _cam(W, H, 40.0)
    means = torch.tensor([[0, 0, 0.005], [0, 0, -2.0], [50.0, 0, 5.0], [0, 0, 5.0], [0, 0, 2e8]], dtype=dt, requires_grad=True)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 5, dtype=dt)
    scales = torch.full((5, 3), 0.2, dtype=dt, requires_grad=True)
    rc, ra, meta = O.rasterization(means, quats, scales, torch.full((5,), 0.5, dtype=dt), torch.rand(5, 3, dtype=dt),
                                   viewmats, Ks, W, H, near_plane=0.01, far_plane=1e8)
    assert meta["radii"][0].tolist()[:3] == [0, 0, 0] and meta["radii"][0, 3] > 0 and meta["radii"][0, 4] == 0
    rc.sum().backward()
    assert means.grad[[0, 1, 2, 4]].abs().max() == 0 and scales.grad[[0, 1, 2, 4]].abs().max() == 0
    assert means.grad[3].abs().max() > 0
    # radius_clip
    _, _, meta2 = O.rasterization(means.detach(), quats, scales.detach(), torch.full((5,), 0.5, dtype=dt),
                                  torch.rand(5, 3, dtype=dt), viewmats, Ks, W, H, radius_clip=100.0)
    assert meta2["radii"].sum() == 0


def test_kat4_alpha_threshold_and_early_stop():
    """alpha < 1/255 is skipped (one pixel each side of the boundary); T <= 1e-4 stops before adding."""
    W = H = 16
    m2 = torch.tensor([[[8.5, 8.5]]], dtype=dt)
    con = torch.tensor([[[1.0, 0.0, 1.0]]], dtype=dt)
    col = torch.ones(1, 1, 1, dtype=dt)
    off = torch.zeros(1, 1, 1, dtype=torch.int32)
    fid = torch.zeros(1, dtype=torch.int32)
    for o, expect in ((1.0 / 255 * 1.0001, True), (1.0 / 255 * 0.9999, False)):
        rc, ra = O.rasterize_to_pixels(m2, con, col, torch.tensor([[o]], dtype=dt), W, H, 16, off, fid)
        assert (ra[0, 8, 8, 0].item() > 0) == expect
    # early stop: 4 opaque splats -> T after k: 1e-3, 1e-6 ...  the 2nd would give T=1e-6 <= 1e-4: NOT added
    n = 4
    m2 = torch.full((1, n, 2), 8.5, dtype=dt)
    con = torch.tensor([[[1.0, 0.0, 1.0]] * n], dtype=dt)
    col = torch.tensor([[[1.0], [10.0], [100.0], [1000.0]]], dtype=dt)
    op = torch.ones(1, n, dtype=dt)
    fid = torch.arange(n, dtype=torch.int32)
    rc, ra, last = O.rasterize_to_pixels(m2, con, col, op, W, H, 16, off, fid, return_last_ids=True)
    assert abs(rc[0, 8, 8, 0].item() - 0.999) < 1e-12 and abs(ra[0, 8, 8, 0].item() - 0.999) < 1e-12
    assert last[0, 8, 8].item() == 0
    rc_c, ra_c, last_c = CO.rasterize_to_pixels(m2, con, col, op, W, H, 16, off, fid, return_last_ids=True)
    assert torch.allclose(rc_c, rc, atol=1e-12) and torch.equal(last_c, last)


def test_kat5_spherical_harmonics():
    """degree 0 inverts rgb_to_sh (reference utils.py:148-150); degrees 1-4 against scipy's complex
    spherical harmonics converted to the real basis with the 3DGS sign convention."""
    from scipy.special import sph_harm
    rgb = torch.rand(7, 3, dtype=dt)
    sh0 = (rgb - 0.5) / 0.28209479177387814
    out = O.spherical_harmonics(0, torch.randn(7, 3, dtype=dt), sh0[:, None, :]) + 0.5
    assert torch.allclose(out, rgb, atol=1e-12)
    g = torch.Generator().manual_seed(0)
    d = torch.randn(64, 3, generator=g, dtype=dt)
    d = d / d.norm(dim=-1, keepdim=True)
    Y = O.eval_sh_bases(4, d).numpy()
    x, y, z = d[:, 0].numpy(), d[:, 1].numpy(), d[:, 2].numpy()
    theta = np.arctan2(y, x)            # azimuth
    phi = np.arccos(np.clip(z, -1, 1))  # polar
    k = 0
    for l in range(5):
        for m in range(-l, l + 1):
            c = sph_harm(abs(m), l, theta, phi)
            if m < 0:
                real = math.sqrt(2) * (-1) ** m * c.imag
            elif m == 0:
                real = c.real
            else:
                real = math.sqrt(2) * (-1) ** m * c.real
            # 3DGS convention = standard real SH (Condon-Shortley phase removed): compare up to that sign
            assert np.allclose(np.abs(Y[:, k]), np.abs(real), atol=1e-10), (l, m)
            ratio = Y[:, k] / np.where(np.abs(real) > 1e-6, real, 1.0)
            sel = np.abs(real) > 1e-6
            assert np.allclose(ratio[sel], ratio[sel][0], atol=1e-8), (l, m)   # one global sign per basis
            k += 1
    # 3DGS signs for l=1: (-y, z, -x)
    assert np.allclose(Y[:, 1], -0.4886025119029199 * y) and np.allclose(Y[:, 2], 0.4886025119029199 * z)
    assert np.allclose(Y[:, 3], -0.4886025119029199 * x)


def test_kat6_tile_binning_hand_computed():
    """radius-17 splat centred on a tile corner (32,32) of a 64x64 image, tile 16: AABB covers
    floor((32-17)/16)=0 .. ceil((32+17)/16)=4 in both axes -> 16 tiles, row-major emission."""
    m2 = torch.tensor([[[32.0, 32.0]]])
    radii = torch.tensor([[17]], dtype=torch.int32)
    dep = torch.tensor([[2.0]])
    tpg, ids, flat = O.isect_tiles(m2, radii, dep, 16, 4, 4, sort=False)
    assert tpg.item() == 16 and flat.tolist() == [0] * 16
    tiles = ((ids >> 32) & 0x1F).tolist()
    assert tiles == list(range(16))
    assert (ids & 0xFFFFFFFF).tolist() == [int(np.float32(2.0).view(np.int32))] * 16
    # radius 15 at the same place: floor(17/16)=1 .. ceil(47/16)=3 -> 2x2 tiles 5,6,9,10
    tpg, ids, flat = O.isect_tiles(m2, torch.tensor([[15]], dtype=torch.int32), dep, 16, 4, 4)
    assert ((ids >> 32) & 0x1F).tolist() == [5, 6, 9, 10]
    off = O.isect_offset_encode(ids, 1, 4, 4).reshape(-1).tolist()
    assert off == [0, 0, 0, 0, 0, 0, 1, 2, 2, 2, 3, 4, 4, 4, 4, 4]
    # partly outside the image: clamped
    tpg, ids, _ = O.isect_tiles(torch.tensor([[[-3.0, 70.0]]]), torch.tensor([[10]], dtype=torch.int32), dep, 16, 4, 4)
    assert tpg.item() == 1 and ((ids >> 32) & 0x1F).tolist() == [12]


@pytest.mark.parametrize("camera_model", ["pinhole", "fisheye", "ortho"])
def test_kat7_gradcheck(camera_model):
    """torch.autograd.gradcheck of the whole float64 path for every differentiable input."""
    W = H = 16
    g = torch.Generator().manual_seed(2)
    N = 6
    means = torch.randn(N, 3, generator=g, dtype=dt) * 0.4 + torch.tensor([0, 0, 4.0], dtype=dt)
    quats = torch.randn(N, 4, generator=g, dtype=dt)
    scales = torch.rand(N, 3, generator=g, dtype=dt) * 0.3 + 0.2
    opac = torch.rand(N, generator=g, dtype=dt) * 0.5 + 0.3
    sh = torch.randn(N, 4, 3, generator=g, dtype=dt) * 0.3
    viewmats, Ks = _cam(W, H, 14.0 if camera_model != "ortho" else 5.0)
    viewmats = viewmats.clone()
    viewmats[0, :3, 3] = torch.tensor([0.1, -0.05, 0.2], dtype=dt)
    ins = [t.requires_grad_() for t in (means, quats, scales, opac, sh, viewmats)]

    def f(means, quats, scales, opac, sh, viewmats):
        rc, ra, _ = O.rasterization(means, quats, scales, opac, sh, viewmats, Ks, W, H, sh_degree=1,
                                    camera_model=camera_model, render_mode="RGB+D")
        return rc, ra

    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-3, nondet_tol=0)


def test_kat8_fisheye_equidistant_mapping():
    """A point 45 degrees off-axis lands f*pi/4 pixels from the principal point."""
    W = H = 200
    f = 50.0
    viewmats, Ks = _cam(W, H, f)
    means = torch.tensor([[1.0, 0.0, 1.0], [0.0, -2.0, 2.0]], dtype=dt)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 2, dtype=dt)
    _, m2, _, _, _ = O.fully_fused_projection(means, None, quats, torch.full((2, 3), 0.05, dtype=dt), viewmats, Ks,
                                              W, H, camera_model="fisheye")
    assert abs(m2[0, 0, 0].item() - (W / 2 + f * math.pi / 4)) < 1e-5 and abs(m2[0, 0, 1].item() - H / 2) < 1e-5
    assert abs(m2[0, 1, 1].item() - (H / 2 - f * math.pi / 4)) < 1e-5


def test_kat9_permutation_invariance_and_kat10_absgrad():
    W = H = 32
    g = torch.Generator().manual_seed(4)
    N = 40
    means = torch.randn(N, 3, generator=g, dtype=dt) * 0.6 + torch.tensor([0, 0, 5.0], dtype=dt)
    quats = torch.randn(N, 4, generator=g, dtype=dt)
    scales = torch.rand(N, 3, generator=g, dtype=dt) * 0.3 + 0.05
    opac = torch.rand(N, generator=g, dtype=dt) * 0.8 + 0.1
    cols = torch.rand(N, 3, generator=g, dtype=dt)
    viewmats, Ks = _cam(W, H, 30.0)
    rc, ra, _ = O.rasterization(means, quats, scales, opac, cols, viewmats, Ks, W, H)
    perm = torch.randperm(N, generator=g)
    rc2, ra2, _ = O.rasterization(means[perm], quats[perm], scales[perm], opac[perm], cols[perm], viewmats, Ks, W, H)
    assert torch.allclose(rc, rc2, atol=1e-12) and torch.allclose(ra, ra2, atol=1e-12)
    # absgrad >= |grad| elementwise, both oracles agree
    probe, absout = [], []
    m = means.clone().requires_grad_()
    rc, ra, meta = O.rasterization(m, quats, scales, opac, cols, viewmats, Ks, W, H, absgrad_probe=probe)
    meta["means2d"].retain_grad()
    (rc * torch.rand(rc.shape, generator=g, dtype=dt)).sum().backward()
    ab = O.collect_absgrad(probe, N).reshape(1, N, 2)
    assert (ab + 1e-15 >= meta["means2d"].grad.abs()).all() and ab.sum() > 0


def _seam_checks():
    """The panorama is periodic in x (build-defined, oracle/torch_oracle.py::isect_tiles periodic=True): a splat straddling
    the +-pi seam (directly BEHIND the camera) continues on the other side of the image, and turning the camera about
    its vertical axis by k columns' worth of longitude rolls the image by k pixels (k a multiple of the tile size, so
    that the 3-sigma tile boxes turn with it)."""
    W, H = 128, 64
    vm, K = _cam(W, H, 50.0)
    means = torch.tensor([[0.0, 0.0, -2.0]], dtype=dt)                 # longitude pi: the seam
    quats = torch.tensor([[1.0, 0, 0, 0]], dtype=dt)
    args = (quats, torch.full((1, 3), 0.3, dtype=dt), torch.tensor([0.9], dtype=dt), torch.tensor([[0.2, 0.6, 0.9]], dtype=dt))
    rc, ra, meta = O.rasterization(means, *args, vm, K, W, H, camera_model="spherical")
    x_seam = float(meta["means2d"][0, 0, 0])
    assert min(x_seam, W - x_seam) < 1e-6 and int(meta["radii"][0, 0]) > 8
    assert int(meta["tiles_per_gauss"][0, 0]) >= 4                      # tile columns on BOTH sides
    a = ra[0, :, :, 0]
    assert float(a[H // 2, 0]) > 0.5 and float(a[H // 2, W - 1]) > 0.5
    assert (a - a.flip(1)).abs().max() < 1e-9                           # an isotropic blob on the seam: mirror symmetric
    # the C restatement takes the same decision
    rc_c, ra_c, _ = O.rasterization(means, *args, vm, K, W, H, camera_model="spherical", raster_fn=CO.raster_fn())
    assert (rc_c - rc).abs().max() < 1e-12 and (ra_c - ra).abs().max() < 1e-12
    # a footprint wider than the image: every tile column once, each at the copy nearest to it -- the render is still
    # periodic, i.e. invariant under a turn of the camera by whole tile columns
    big = torch.tensor([[0.3, 0.9, -0.6]], dtype=dt)
    bargs = (quats, torch.tensor([[1.5, 0.2, 0.2]], dtype=dt), torch.tensor([0.8], dtype=dt), torch.tensor([[0.9, 0.5, 0.1]], dtype=dt))
    rc_w, ra_w, meta_w = O.rasterization(big, *bargs, vm, K, W, H, camera_model="spherical")
    assert int(meta_w["radii"][0, 0]) > W // 2 and int(meta_w["tiles_per_gauss"][0, 0]) % (W // 16) == 0
    th2 = 2.0 * math.pi * 32 / W
    Ry2 = torch.tensor([[math.cos(th2), 0.0, math.sin(th2), 0.0], [0.0, 1.0, 0.0, 0.0],
                        [-math.sin(th2), 0.0, math.cos(th2), 0.0], [0.0, 0.0, 0.0, 1.0]], dtype=dt)
    rc_w2, _, _ = O.rasterization(big, *bargs, Ry2[None] @ vm, K, W, H, camera_model="spherical", raster_fn=CO.raster_fn())
    assert (torch.roll(rc_w, 32, dims=2) - rc_w2).abs().max() < 1e-9
    # when the tile grid does not line up (W % tile_size != 0) the image is not periodic: the blob is cut at the edge
    W2 = 120
    _, ra2, meta2 = O.rasterization(means, *args, vm, K, W2, H, camera_model="spherical")
    a2 = ra2[0, :, :, 0]
    assert min(float(a2[H // 2, 0]), float(a2[H // 2, W2 - 1])) == 0.0
    # yaw <-> roll, with Gaussians all around and gradients
    g = torch.Generator().manual_seed(21)
    N = 40
    d = torch.randn(N, 3, generator=g, dtype=dt)
    d[:, 1] *= 0.5
    means = (d / d.norm(dim=-1, keepdim=True) * (1.5 + torch.rand(N, 1, generator=g, dtype=dt))).requires_grad_()
    qs = torch.randn(N, 4, generator=g, dtype=dt)
    scales = torch.rand(N, 3, generator=g, dtype=dt) * 0.3 + 0.05
    opac = torch.rand(N, generator=g, dtype=dt) * 0.6 + 0.2
    cols = torch.rand(N, 3, generator=g, dtype=dt)
    k = 48                                                              # three tile columns
    th = 2.0 * math.pi * k / W
    # x = W/2 + W lon / (2 pi), lon = atan2(x_cam, z_cam): adding th to every longitude = turning the camera by -th
    Ry = torch.tensor([[math.cos(th), 0.0, math.sin(th), 0.0], [0.0, 1.0, 0.0, 0.0],
                       [-math.sin(th), 0.0, math.cos(th), 0.0], [0.0, 0.0, 0.0, 1.0]], dtype=dt)
    wgt = torch.rand(1, H, W, 3, generator=g, dtype=dt)
    out = []
    for view, w in ((vm, wgt), (Ry[None] @ vm, torch.roll(wgt, k, dims=2))):
        rc, ra, meta = O.rasterization(means, qs, scales, opac, cols, view, K, W, H, camera_model="spherical",
                                       raster_fn=CO.raster_fn())
        (gm,) = torch.autograd.grad((rc * w).sum(), means)
        out.append((rc, gm, meta))
    dx = (out[1][2]["means2d"][0, :, 0] - out[0][2]["means2d"][0, :, 0] - k) / W
    assert (dx - dx.round()).abs().max() < 1e-9                         # every centre moved by k columns (mod W)
    assert (torch.roll(out[0][0], k, dims=2) - out[1][0]).abs().max() < 1e-9
    assert (out[0][1] - out[1][1]).abs().max() < 1e-9 * max(1.0, float(out[0][1].abs().max()))
    assert int((out[0][2]["radii"] > 0).sum()) == N


def test_kat11_spherical_mapping_and_gradcheck():
    """The 360-degree model as this build defines it (oracle/torch_oracle.py::_spherical_proj; the fork's kernel is absent
    from the reference): longitude / latitude of the camera-space direction map linearly onto the W x H panorama, depth
    is the range, Gaussians behind the camera are seen; the whole float64 path passes gradcheck with Gaussians all around."""
    W, H = 400, 200
    viewmats, Ks = _cam(W, H, 50.0)                       # K is not used by this model
    means = torch.tensor([[0.0, 0.0, 2.0], [2.0, 0.0, 0.0], [-2.0, 0.0, 0.0], [0.0, 2.0, 2.0], [1.0, 1.0, -1.0]], dtype=dt)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 5, dtype=dt)
    radii, m2, dep, _, _ = O.fully_fused_projection(means, None, quats, torch.full((5, 3), 0.05, dtype=dt), viewmats, Ks,
                                                    W, H, camera_model="spherical")
    assert (radii > 0).all()
    want = torch.tensor([[200.0, 100.0], [300.0, 100.0], [100.0, 100.0], [200.0, 150.0],
                         [200.0 + 400.0 * 135.0 / 360.0, 100.0 + 200.0 * math.degrees(math.atan2(1.0, math.sqrt(2.0))) / 180.0]],
                        dtype=dt)
    assert (m2[0] - want).abs().max() < 1e-4
    assert (dep[0] - means.norm(dim=-1)).abs().max() < 1e-12
    # near / far apply to the range: a Gaussian behind the camera at range 1.7 survives near = 1.5 and not near = 1.8
    for near, seen in ((1.5, True), (1.8, False)):
        r, *_ = O.fully_fused_projection(means[4:], None, quats[4:], torch.full((1, 3), 0.05, dtype=dt), viewmats, Ks, W, H,
                                         near_plane=near, camera_model="spherical")
        assert bool(r[0, 0] > 0) == seen
    Wg, Hg = 32, 16
    g = torch.Generator().manual_seed(6)
    N = 6
    d = torch.randn(N, 3, generator=g, dtype=dt)
    d[:, 1] *= 0.4                                        # away from the poles
    means = d / d.norm(dim=-1, keepdim=True) * (1.5 + torch.rand(N, 1, generator=g, dtype=dt))
    qs = torch.randn(N, 4, generator=g, dtype=dt)
    scales = torch.rand(N, 3, generator=g, dtype=dt) * 0.4 + 0.3
    opac = torch.rand(N, generator=g, dtype=dt) * 0.5 + 0.3
    sh = torch.randn(N, 4, 3, generator=g, dtype=dt) * 0.3
    vm, Kg = _cam(Wg, Hg, 14.0)
    vm = vm.clone()
    vm[0, :3, 3] = torch.tensor([0.1, -0.05, 0.2], dtype=dt)
    ins = [t.requires_grad_() for t in (means, qs, scales, opac, sh, vm)]

    def f(means, qs, scales, opac, sh, vm):
        rc, ra, meta = O.rasterization(means, qs, scales, opac, sh, vm, Kg, Wg, Hg, sh_degree=1,
                                       camera_model="spherical", render_mode="RGB+D")
        return rc, ra

    rc, ra, meta = O.rasterization(*[t.detach() for t in ins], Kg, Wg, Hg, sh_degree=1, camera_model="spherical")
    assert (meta["radii"] > 0).all() and float(ra.max()) > 0.2
    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-3, nondet_tol=0)
