"""Pins the plain-C rasteriser oracle (oracle/c/raster_oracle.c: forward + hand-written backward)
against the pure-torch oracle, whose gradients come from autograd."""
import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as O
from tests.util import small_scene, two_cameras


@pytest.mark.parametrize("with_bg,sh_degree,mode", [(True, 3, "RGB"), (False, 1, "RGB+ED")])
def test_c_oracle_matches_torch_oracle(with_bg, sh_degree, mode):
    W, H, N = 80, 56, 1500
    means, quats, scales, opac, sh = small_scene(N, scale=0.25)
    viewmats, Ks = two_cameras(W, H)
    X = 3 if mode == "RGB" else 4
    bg = torch.rand(2, 3) if with_bg else None
    g = torch.Generator().manual_seed(5)
    wc = torch.rand(2, H, W, X, generator=g, dtype=torch.float64)
    wa = torch.rand(2, H, W, 1, generator=g, dtype=torch.float64)
    res = []
    for use_c in (False, True):
        ps = [t.clone().double().requires_grad_() for t in (means, quats, scales, opac, sh)]
        probe, absout = [], []
        kw = dict(raster_fn=CO.raster_fn(absout)) if use_c else dict(absgrad_probe=probe)
        rc, ra, meta = O.rasterization(*ps, viewmats, Ks, W, H, sh_degree=sh_degree, backgrounds=bg, render_mode=mode, **kw)
        meta["means2d"].retain_grad()
        ((rc * wc).sum() + (ra * wa).sum()).backward()
        ab = absout[0] if use_c else O.collect_absgrad(probe, 2 * N).reshape(2, N, 2)
        res.append((rc.detach(), ra.detach(), [p.grad for p in ps], meta["means2d"].grad, ab))
    a, b = res
    assert (a[0] - b[0]).abs().max() < 1e-12 and (a[1] - b[1]).abs().max() < 1e-12
    for x, y in zip(a[2], b[2]):
        assert ((x - y).norm() / x.norm()) < 1e-10
    assert ((a[3] - b[3]).norm() / a[3].norm()) < 1e-10
    assert ((a[4] - b[4]).norm() / a[4].norm()) < 1e-10


def test_c_oracle_f32_build_and_threads():
    assert CO._lib(torch.float32).oracle_real_size() == 4
    assert CO.max_threads() >= 1
    # f32 build runs the same scene within float tolerance of the f64 build
    W, H, N = 64, 48, 800
    means, quats, scales, opac, sh = small_scene(N, scale=0.3)
    viewmats, Ks = two_cameras(W, H)
    outs = []
    for dtype in (torch.float64, torch.float32):
        rc, ra, _ = O.rasterization(means, quats, scales, opac, sh, viewmats, Ks, W, H, sh_degree=2,
                                    raster_fn=CO.raster_fn(), dtype=dtype)
        outs.append(rc.double())
    assert (outs[0] - outs[1]).abs().mean() < 1e-5
