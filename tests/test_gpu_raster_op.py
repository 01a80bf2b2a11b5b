"""`rasterization()` as ONE library call each way (splat_one_amd/raster_op.py: so_rasterization_fwd / _bwd) -- the path the
reference's own call shape takes (/root/reference/utils/gsplat_utils/gsplat_trainer.py:477-494): it must return what the
operator-by-operator composition returns (images, alphas, every `meta` entry, every gradient, `info["means2d"].grad` /
`.absgrad` for the strategy hooks :616-622, :744-752), and what the float64 oracle returns (tests/test_gpu_rasterization.py
runs every RGB case through it already).  Here: the two product paths against each other, the index outputs bit for bit
against the oracle, re-entrancy, and the bin sizing that replaces the host read of the intersection count."""
import threading
import warnings

import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as O
from splat_one_amd.scene import make_scene

pytestmark = pytest.mark.gpu


def _inputs(splats, to, grad=True):
    p = {k: v.detach().clone().to(to).requires_grad_(grad) for k, v in splats.items()}
    return p, (p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), torch.cat([p["sh0"], p["shN"]], 1))


def _step(fn, splats, viewmats, Ks, W, H, w_rgb, w_a, to, **kw):
    p, args = _inputs(splats, to)
    bg = kw.pop("backgrounds", None)
    if bg is not None:
        bg = bg.detach().clone().to(to).requires_grad_(True)
    rc, ra, meta = fn(*args, viewmats.to(to), Ks.to(to), W, H, near_plane=0.01, far_plane=1e8, packed=False, backgrounds=bg, **kw)
    meta["means2d"].retain_grad()
    ((rc * w_rgb.to(rc)).sum() + (ra * w_a.to(ra)).sum()).backward()
    g = {k: v.grad.detach().cpu().double() for k, v in p.items()}
    g["means2d"] = meta["means2d"].grad.detach().cpu().double()
    if kw.get("absgrad"):
        g["absgrad"] = meta["means2d"].absgrad.detach().cpu().double()
    if bg is not None:
        g["backgrounds"] = bg.grad.detach().cpu().double()
    return rc.detach().cpu().double(), ra.detach().cpu().double(), g, meta


@pytest.mark.parametrize("camera_model,C,aa,absgrad,bg", [("pinhole", 1, False, False, False), ("pinhole", 3, True, True, True),
                                                            ("fisheye", 2, False, True, False), ("spherical", 1, False, False, True),
                                                            ("ortho", 1, True, False, False)])
def test_one_call_equals_the_operator_composition(dev, camera_model, C, aa, absgrad, bg):
    from splat_one_amd import rasterization
    W, H = (128, 64) if camera_model == "spherical" else (150, 97)
    N = 4000
    splats, c2w, Ks = make_scene(N, W, H, regime="ref", n_views=C)
    g = torch.Generator().manual_seed(5)
    splats["scales"] = splats["scales"] + torch.randn(N, 3, generator=g) * 0.4
    if camera_model == "spherical":
        c2w = torch.eye(4)[None].repeat(C, 1, 1)
    if camera_model == "ortho":
        Ks = Ks.clone()
        Ks[:, 0, 0], Ks[:, 1, 1] = 12.0, 11.0
    viewmats = torch.linalg.inv(c2w)
    w_rgb, w_a = torch.rand(C, H, W, 3, generator=g), torch.rand(C, H, W, 1, generator=g)
    kw = dict(sh_degree=3, camera_model=camera_model, rasterize_mode="antialiased" if aa else "classic", absgrad=absgrad,
              backgrounds=torch.rand(C, 3, generator=g) if bg else None)
    rc1, ra1, g1, m1 = _step(rasterization, splats, viewmats, Ks, W, H, w_rgb, w_a, dev, fused=True, **kw)
    rc0, ra0, g0, m0 = _step(rasterization, splats, viewmats, Ks, W, H, w_rgb, w_a, dev, fused=False, **kw)
    # the same arithmetic (csrc/splat_math.hpp) compiled into different kernels: images agree to rounding -- except where a
    # one-ulp difference of a conic moves a pixel across the alpha >= 1/255 threshold (4e-3 x colour, a handful of pixels)
    for a, b in ((rc1, rc0), (ra1, ra0)):
        d = (a - b).abs()
        assert d.mean().item() <= 5e-7 and int((d > 1e-5).sum()) <= 1e-4 * d.numel() and d.max().item() <= 5e-3, (d.mean(), d.max())
    for k in g0:
        err, ref = (g1[k] - g0[k]).norm().item(), g0[k].norm().item()
        assert err <= 1e-4 * ref + 1e-9, (k, err, ref)            # float atomics in another order, the odd threshold pixel
    for k in ("radii", "means2d", "depths", "conics", "opacities"):
        a, b = m1[k].detach().cpu(), m0[k].detach().cpu()
        assert a.shape == b.shape and a.dtype == b.dtype, k
        vis = (m0["radii"].cpu() > 0)
        assert torch.equal(m1["radii"].cpu(), m0["radii"].cpu())
        if k != "radii":
            assert (a[vis].double() - b[vis].double()).abs().max().item() <= 1e-5 * max(1.0, b[vis].abs().max().item()), k
    # index outputs: gsplat's lists, bit for bit, from both paths (tile_cull only shortens the kernels' private lists)
    same_depth_bits = torch.equal(m1["depths"].detach(), m0["depths"].detach())      # (range depth: one more sqrt, a last-bit matter)
    for k in ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert k in m1
        if k == "isect_ids" and not same_depth_bits:
            assert torch.equal(m1[k] >> 32, m0[k] >> 32)         # camera | tile part of the keys
        elif k != "flatten_ids" or same_depth_bits:
            assert torch.equal(m1[k], m0[k]), k
    assert int(m1["n_isects_kernel"]) <= m1["flatten_ids"].numel()
    for k in ("width", "height", "tile_size", "n_cameras", "tile_width", "tile_height"):
        assert m1[k] == m0[k], k


def test_index_outputs_equal_the_oracles_lists(dev):
    """meta["isect_ids" / "flatten_ids" / "tiles_per_gauss" / "isect_offsets"] of the default call == the float64 oracle's
    (gsplat's global stable sort by camera | tile | depth bits) when both see the same float32 projections."""
    from splat_one_amd import rasterization
    W, H, N, C = 200, 120, 5000, 2
    splats, c2w, Ks = make_scene(N, W, H, regime="ref", n_views=C)
    viewmats = torch.linalg.inv(c2w)
    _, args = _inputs(splats, dev, grad=False)
    with torch.no_grad():
        _, _, m = rasterization(*args, viewmats.to(dev), Ks.to(dev), W, H, sh_degree=1, packed=False)
    tpg, ids, flat = O.isect_tiles(m["means2d"].cpu().double(), m["radii"].cpu(), m["depths"].cpu().double(), 16, 13, 8)
    assert torch.equal(m["tiles_per_gauss"].cpu(), tpg.to(torch.int32))
    assert torch.equal(m["flatten_ids"].cpu().long(), flat.long()) and torch.equal(m["isect_ids"].cpu(), ids)
    assert torch.equal(m["isect_offsets"].cpu().long(), O.isect_offset_encode(ids, C, 13, 8).long())


def test_two_forwards_then_two_backwards(dev):
    """nothing a call saves for its backward is shared with a later call (per-call records, lists and counters)"""
    from splat_one_amd import rasterization
    W, H, N = 96, 80, 3000
    splats, c2w, Ks = make_scene(N, W, H, regime="ref", n_views=2)
    vm = torch.linalg.inv(c2w).to(dev)
    g = torch.Generator().manual_seed(1)
    w = torch.rand(1, H, W, 3, generator=g).to(dev)

    def fwd(view, p):
        args = (p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), torch.cat([p["sh0"], p["shN"]], 1))
        rc, _, meta = rasterization(*args, vm[view:view + 1], Ks[view:view + 1].to(dev), W, H, sh_degree=3, packed=False)
        return (rc * w).sum(), meta

    pa, _ = _inputs(splats, dev)
    la, _ = fwd(0, pa)
    la.backward()
    ref0 = {k: v.grad.clone() for k, v in pa.items()}
    pb, _ = _inputs(splats, dev)
    lb, _ = fwd(1, pb)
    lb.backward()
    ref1 = {k: v.grad.clone() for k, v in pb.items()}
    pc, _ = _inputs(splats, dev)
    l0, m0 = fwd(0, pc)
    l1, m1 = fwd(1, pc)                  # second forward before the first backward
    m0["means2d"].retain_grad()
    l0.backward()
    g0 = {k: v.grad.clone() for k, v in pc.items()}
    l1.backward()
    for k in g0:
        # (isotropic splats: the quaternion gradient is rounding noise of terms the size of the scale gradient)
        floor = 1e-5 * ref0["scales"].norm() if k == "quats" else 1e-9
        assert (g0[k] - ref0[k]).norm() <= 2e-5 * ref0[k].norm() + floor, k
        assert (pc[k].grad - (ref0[k] + ref1[k])).norm() <= 2e-5 * (ref0[k] + ref1[k]).norm() + 2 * floor, k
    assert m0["means2d"].grad is not None and m0["means2d"].grad.shape == (1, N, 2)


def test_no_grad_and_threads(dev):
    """forward-only calls save nothing; two Python threads render at once (training thread + GUI thread,
    app/gsplat_manager.py:185 vs :204-206) and get their own images"""
    from splat_one_amd import rasterization
    W, H, N = 160, 96, 6000
    splats, c2w, Ks = make_scene(N, W, H, regime="ref", n_views=2)
    vm = torch.linalg.inv(c2w).to(dev)
    _, args = _inputs(splats, dev, grad=False)
    with torch.no_grad():       # (lists are sorted before they are walked: a render is bit-reproducible)
        want = [rasterization(*args, vm[v:v + 1], Ks[v:v + 1].to(dev), W, H, sh_degree=3, packed=False)[0].clone() for v in range(2)]
        slow = [rasterization(*args, vm[v:v + 1], Ks[v:v + 1].to(dev), W, H, sh_degree=3, packed=False, fused=False)[0] for v in range(2)]
    for v in range(2):
        assert (want[v] - slow[v]).abs().mean().item() <= 5e-7
    got = [[None] * 20, [None] * 20]

    def work(v):
        with torch.no_grad():
            for i in range(20):
                got[v][i] = rasterization(*args, vm[v:v + 1], Ks[v:v + 1].to(dev), W, H, sh_degree=3, packed=False)[0]

    ts = [threading.Thread(target=work, args=(v,)) for v in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    for v in range(2):
        for im in got[v]:
            assert torch.equal(im, want[v])


def test_bins_follow_the_scene_without_host_reads(dev, monkeypatch):
    """The slot count per tile is measured on the first call of a tile grid (and again when the model has grown 1.5x) and
    then followed through the host-mapped status word.  What an overflow can do (raster_op.py docstring): a call
    without gradient waits for its own status and runs again -- never an image from cut lists; a differentiated call does
    not wait: its backward returns EXACTLY zero gradients (decided on the device), `pending_overflow()` and the next call
    say so, and the bins are repaired."""
    from splat_one_amd import raster_op, rasterization
    W, H = 112, 80                     # a tile grid no other test uses: fresh bin state
    raster_op._BINS.clear()
    monkeypatch.setattr(raster_op, "_MIN_SLOTS", 32)
    splats, c2w, Ks = make_scene(3000, W, H, regime="ref")
    vm, Kd = torch.linalg.inv(c2w).to(dev), Ks.to(dev)
    _, args = _inputs(splats, dev, grad=False)

    def render(a):
        with torch.no_grad():
            return rasterization(*a, vm, Kd, W, H, sh_degree=3, packed=False)[0]

    im = render(args)                  # first call: probes, finds 32 slots too few, reruns on 8x the fullest tile
    ref = im.clone()
    slow = rasterization(*args, vm, Kd, W, H, sh_degree=3, packed=False, fused=False)[0]
    (bins,) = [b for k, b in raster_op._BINS.items() if k[2] == 7 * 5]
    assert bins.probed and bins.n_probe == 3000 and bins.slots > 32 and (im - slow).abs().mean().item() <= 5e-7
    torch.cuda.synchronize()
    fullest = int(bins.status[0])
    # (1) no gradient wanted: bins shrunk behind the module's back -> the call notices by itself and returns the exact image
    bins.slots = max(16, fullest // 4)
    bins.status[2] = -1                # (as if the previous call were still in flight: nothing to look at before this one)
    assert torch.equal(render(args), ref) and bins.slots >= fullest and raster_op.pending_overflow() == 0
    # (2) a differentiated call on bins that are too small: not waited for; zero gradients, announced afterwards
    leaves, gargs = _inputs(splats, dev, grad=True)
    bins.slots = max(16, fullest // 4)
    bins.status[2] = -1
    rc, _ra, info = rasterization(*gargs, vm, Kd, W, H, sh_degree=3, packed=False)
    info["means2d"].retain_grad()
    (rc * rc).sum().backward()
    torch.cuda.synchronize()
    assert int(bins.status[1]) == 1 and (rc.detach() - ref).abs().max().item() > 1e-3       # that image WAS cut ...
    for t in leaves.values():                                                                # ... and taught nothing
        assert t.grad is not None and float(t.grad.abs().max()) == 0.0
    assert float(info["means2d"].grad.abs().max()) == 0.0
    assert raster_op.pending_overflow() == 1
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        leaves, gargs = _inputs(splats, dev, grad=True)
        rc2, _, _ = rasterization(*gargs, vm, Kd, W, H, sh_degree=3, packed=False)
        (rc2 * rc2).sum().backward()
    assert any("list slots" in str(w.message) for w in wl)                                   # the next call says so ...
    assert bins.slots >= 4 * fullest and torch.equal(rc2.detach(), ref)                      # ... on repaired bins
    assert all(float(t.grad.abs().max()) > 0.0 for t in leaves.values()) and raster_op.pending_overflow() == 0
    # (3) growth without overflow: three times the Gaussians at the same resolution -> measured again at once (N > 1.5x)
    big, _, _ = make_scene(9000, W, H, regime="ref")
    leaves_big, args_big = _inputs(big, dev, grad=True)
    bins.slots = int(1.5 * fullest)    # tight but sufficient for the small scene ...
    bins.status[2] = -1
    assert torch.equal(render(args), ref)
    out_big, _, _ = rasterization(*args_big, vm, Kd, W, H, sh_degree=3, packed=False)       # ... the big one is re-probed
    torch.cuda.synchronize()
    assert bins.n_probe == 9000 and int(bins.status[1]) == 0 and bins.slots >= 2 * int(bins.status[0])
    with torch.no_grad():
        slow_big = rasterization(*args_big, vm, Kd, W, H, sh_degree=3, packed=False, fused=False)[0]
    assert (out_big.detach() - slow_big).abs().mean().item() <= 5e-7


def test_raw_parameter_call_equals_the_composed_call(dev):
    """`rasterization_from_parameters` (what splat_one_amd's Runner.rasterize_splats calls: activations inside the kernels)
    == rasterization(means, quats, exp(.), sigmoid(.), cat(.)) as the reference composes it at gsplat_trainer.py:456-494:
    images, alphas, info, and the gradients of the RAW parameters."""
    from splat_one_amd.rendering import rasterization, rasterization_from_parameters
    W, H, N, C = 144, 90, 5000, 2
    splats, c2w, Ks = make_scene(N, W, H, regime="ref", n_views=C)
    g = torch.Generator().manual_seed(6)
    splats["scales"] = splats["scales"] + torch.randn(N, 3, generator=g) * 0.4
    vm, Kd = torch.linalg.inv(c2w).to(dev), Ks.to(dev)
    w_rgb, w_a = torch.rand(C, H, W, 3, generator=g).to(dev), torch.rand(C, H, W, 1, generator=g).to(dev)
    res = []
    for raw in (True, False):
        p = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in splats.items()}
        kw = dict(sh_degree=3, near_plane=0.01, far_plane=1e8, packed=False, absgrad=True, rasterize_mode="antialiased")
        if raw:
            rc, ra, meta = rasterization_from_parameters(p["means"], p["quats"], p["scales"], p["opacities"], p["sh0"], p["shN"],
                                                         vm, Kd, W, H, **kw)
        else:
            rc, ra, meta = rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                         torch.cat([p["sh0"], p["shN"]], 1), vm, Kd, W, H, **kw)
        meta["means2d"].retain_grad()
        ((rc * w_rgb).sum() + (ra * w_a).sum()).backward()
        res.append((rc.detach(), ra.detach(), {k: v.grad.clone() for k, v in p.items()}, meta))
    (rc1, ra1, g1, m1), (rc0, ra0, g0, m0) = res
    assert (rc1 - rc0).abs().mean().item() <= 5e-7 and (ra1 - ra0).abs().mean().item() <= 5e-7
    for k in g0:
        assert g1[k].shape == g0[k].shape and (g1[k] - g0[k]).norm() <= 1e-4 * g0[k].norm() + 1e-9, k
    for a, b in ((m1["means2d"].grad, m0["means2d"].grad), (m1["means2d"].absgrad, m0["means2d"].absgrad)):
        assert (a - b).norm() <= 1e-4 * b.norm()
    assert torch.equal(m1["radii"], m0["radii"]) and torch.equal(m1["tiles_per_gauss"], m0["tiles_per_gauss"])
    # a shape outside the common one (packed) is composed exactly as the reference composes it
    with torch.no_grad():
        a = rasterization_from_parameters(*[splats[k].to(dev) for k in ("means", "quats", "scales", "opacities", "sh0", "shN")],
                                          vm, Kd, W, H, sh_degree=3, packed=True)[0]
        b = rasterization(splats["means"].to(dev), splats["quats"].to(dev), splats["scales"].to(dev).exp(), splats["opacities"].to(dev).sigmoid(),
                          torch.cat([splats["sh0"], splats["shN"]], 1).to(dev), vm, Kd, W, H, sh_degree=3, packed=True)[0]
    assert torch.equal(a, b)


def test_dense_lists_take_the_tile_wave_backward(dev):
    """Long tile lists (dense initialisations): `rasterization()` switches its backward to the one-wave-per-tile kernel
    (so_raster_desc.raster_impl = 1) from the mean list length its bins' status word reports -- no host read -- and the
    gradients stay those of the operator composition."""
    from splat_one_amd import raster_op, rasterization
    W, H, N = 144, 112, 40_000                    # a tile grid no other test uses: 9 x 7 tiles, several hundred entries each
    splats, c2w, Ks = make_scene(N, W, H, regime="ref")
    vm, Kd = torch.linalg.inv(c2w).to(dev), Ks.to(dev)
    g = torch.Generator().manual_seed(4)
    splats = {k: v.detach().clone() for k, v in splats.items()}
    splats["scales"] = splats["scales"] + torch.randn(N, 3, generator=g) * 0.4      # anisotropic: the quaternion gradient is a signal
    w_rgb, w_a = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 1, generator=g)
    ref = _step(rasterization, splats, vm, Kd, W, H, w_rgb, w_a, dev, sh_degree=3, fused=False)
    out = None
    for _ in range(2):                             # the second call has looked at the first one's status word
        out = _step(rasterization, splats, vm, Kd, W, H, w_rgb, w_a, dev, sh_degree=3)
    (bins,) = [b for k, b in raster_op._BINS.items() if k[2] == 9 * 7]
    assert bins.mean_list >= 256.0, bins.mean_list
    assert (out[0] - ref[0]).abs().max().item() <= 2e-6
    for k in ref[2]:
        floor = 1e-5 * ref[2]["scales"].norm() if k == "quats" else 0.0
        assert ((out[2][k] - ref[2][k]).norm() / (ref[2][k].norm() + floor)).item() <= 2e-5, k
