"""CPU check of the PRODUCT's per-Gaussian math header (splat_one_amd/csrc/splat_math.hpp) built
with g++ by tests/host_harness -- the same code the gfx950 kernels inline -- against the float64
autograd oracle: projection forward/backward for every camera model (quat/scale and covariance
inputs, compensation, view-matrix gradients) and the SH bases with their derivatives."""
import ctypes

import pytest
import torch

from oracle import torch_oracle as O
from tests.host_harness.build import build
from tests.util import rel_err, small_scene, two_cameras

dt = torch.float64


def p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


@pytest.fixture(scope="module")
def hh():
    return ctypes.CDLL(build())


@pytest.mark.parametrize("model_i,model", [(0, "pinhole"), (1, "ortho"), (2, "fisheye"), (3, "spherical")])
@pytest.mark.parametrize("use_cov", [False, True])
def test_projection_math_f64(hh, model_i, model, use_cov):
    W, H, N = 96, 64, 600
    means, quats, scales, _, _ = (t.double() for t in small_scene(N))
    viewmats, Ks = two_cameras(W, H)
    viewmats, Ks = viewmats.double().contiguous(), Ks.double().contiguous()
    if model == "ortho":
        Ks[:, 0, 0], Ks[:, 1, 1] = 12.0, 11.0
    C = 2
    g = torch.Generator().manual_seed(3)
    wm, wd, wc, wp = (torch.randn(C, N, 2, generator=g, dtype=dt), torch.randn(C, N, generator=g, dtype=dt),
                      torch.randn(C, N, 3, generator=g, dtype=dt), torch.randn(C, N, generator=g, dtype=dt))
    m = means.clone().requires_grad_()
    q = quats.clone().requires_grad_()
    s = scales.clone().requires_grad_()
    V = viewmats.clone().requires_grad_()
    cov = O.quat_scale_to_covar(quats, scales).clone().requires_grad_() if use_cov else None
    radii, m2, dep, con, comp = O.fully_fused_projection(m, cov, None if use_cov else q, None if use_cov else s, V, Ks,
                                                         W, H, calc_compensations=True, camera_model=model,
                                                         near_plane=0.01, far_plane=100.0)
    ((m2 * wm).sum() + (dep * wd).sum() + (con * wc).sum() + (comp * wp).sum()).backward()
    cov6 = None
    if use_cov:
        cd = cov.detach()
        cov6 = torch.stack([cd[:, 0, 0], cd[:, 0, 1], cd[:, 0, 2], cd[:, 1, 1], cd[:, 1, 2], cd[:, 2, 2]], -1).contiguous()
    r2 = torch.zeros(C, N, dtype=torch.int32)
    o_m2, o_d, o_c, o_p = (torch.zeros(C, N, 2, dtype=dt), torch.zeros(C, N, dtype=dt),
                           torch.zeros(C, N, 3, dtype=dt), torch.zeros(C, N, dtype=dt))
    hh.hh_proj_fwd_f64(C, N, p(means), p(cov6), p(None if use_cov else quats), p(None if use_cov else scales),
                       p(viewmats), p(Ks), W, H, ctypes.c_double(0.3), ctypes.c_double(0.01), ctypes.c_double(100.0),
                       ctypes.c_double(0.0), model_i, p(r2), p(o_m2), p(o_d), p(o_c), p(o_p))
    assert (radii > 0).sum() > 300 and torch.equal(r2, radii)
    assert (o_m2 - m2).abs().max() < 1e-10 and (o_d - dep).abs().max() < 1e-12
    assert rel_err(o_c, con) < 1e-12 and (o_p - comp).abs().max() < 1e-12
    vm, vq, vs = torch.zeros(N, 3, dtype=dt), torch.zeros(N, 4, dtype=dt), torch.zeros(N, 3, dtype=dt)
    vc6, vV = torch.zeros(N, 6, dtype=dt), torch.zeros(C, 4, 4, dtype=dt)
    hh.hh_proj_bwd_f64(C, N, p(means), p(cov6), p(None if use_cov else quats), p(None if use_cov else scales),
                       p(viewmats), p(Ks), W, H, ctypes.c_double(0.3), model_i, p(r2), p(wm), p(wd), p(wc), p(wp),
                       p(vm), p(vc6 if use_cov else None), p(None if use_cov else vq), p(None if use_cov else vs), p(vV))
    assert rel_err(vm, m.grad) < 1e-11
    assert rel_err(vV[:, :3, :], V.grad[:, :3, :]) < 1e-11
    if use_cov:
        cg = cov.grad
        cg6 = torch.stack([cg[:, 0, 0], cg[:, 0, 1] + cg[:, 1, 0], cg[:, 0, 2] + cg[:, 2, 0], cg[:, 1, 1],
                           cg[:, 1, 2] + cg[:, 2, 1], cg[:, 2, 2]], -1)
        assert rel_err(vc6, cg6) < 1e-11
    else:
        assert rel_err(vq, q.grad) < 1e-11 and rel_err(vs, s.grad) < 1e-11


def test_projection_math_f32_is_within_bar(hh):
    """The float instantiation (what the kernels run) stays within the 1e-3 gradient bar."""
    W, H, N = 96, 64, 600
    means, quats, scales, _, _ = small_scene(N)
    viewmats, Ks = two_cameras(W, H)
    C = 2
    g = torch.Generator().manual_seed(3)
    wm, wd, wc = torch.randn(C, N, 2, generator=g), torch.randn(C, N, generator=g), torch.randn(C, N, 3, generator=g)
    m, q, s = (t.clone().requires_grad_() for t in (means, quats, scales))
    radii, m2, dep, con, _ = O.fully_fused_projection(m, None, q, s, viewmats, Ks, W, H, near_plane=0.01, far_plane=100.0)
    ((m2 * wm).sum() + (dep * wd).sum() + (con * wc).sum()).backward()
    r2 = radii.clone()
    vm, vq, vs = torch.zeros(N, 3), torch.zeros(N, 4), torch.zeros(N, 3)
    hh.hh_proj_bwd_f32(C, N, p(means), p(None), p(quats), p(scales), p(viewmats.contiguous()), p(Ks.contiguous()), W, H,
                       ctypes.c_double(0.3), 0, p(r2), p(wm), p(wd), p(wc), p(None), p(vm), p(None), p(vq), p(vs), p(None))
    assert rel_err(vm, m.grad) < 1e-4 and rel_err(vq, q.grad) < 1e-3 and rel_err(vs, s.grad) < 1e-4


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_bases_and_derivatives(hh, deg):
    g = torch.Generator().manual_seed(deg)
    d = torch.randn(50, 3, dtype=dt, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    nb = (deg + 1) ** 2
    Y, dY = torch.zeros(50, nb, dtype=dt), torch.zeros(50, nb, 3, dtype=dt)
    hh.hh_sh_bases_f64(deg, ctypes.c_int64(50), p(d), p(Y), p(dY))
    dd = d.clone().requires_grad_()
    Yo = O.eval_sh_bases(deg, dd)
    J = torch.stack([torch.autograd.grad(Yo[:, k].sum() + 0 * dd.sum(), dd, retain_graph=True)[0] for k in range(nb)], 1)
    assert (Y - Yo).abs().max() < 1e-14 and (dY - J).abs().max() < 1e-13


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_eval_sh_bases_fast_matches_the_oracle_basis(deg):
    """`_eval_sh_bases_fast` (the import of the reference's appearance module, utils.py:91, 107): the same real SH basis
    as the oracle's on unit directions, differentiable."""
    from splat_one_amd.ops import _eval_sh_bases_fast
    g = torch.Generator().manual_seed(deg)
    d = torch.randn(500, 3, generator=g, dtype=torch.float64)
    d = (d / d.norm(dim=-1, keepdim=True)).requires_grad_()
    got = _eval_sh_bases_fast((deg + 1) ** 2, d)
    want = O.eval_sh_bases(deg, d.detach())
    assert got.shape == (500, (deg + 1) ** 2) and (got.detach() - want).abs().max() < 1e-14
    with pytest.raises(AssertionError):
        _eval_sh_bases_fast(5, d)
    if deg == 0:
        return                                   # a constant: nothing to differentiate
    w = torch.randn(got.shape, generator=g, dtype=torch.float64)
    d2 = d.detach().clone().requires_grad_()
    (got * w).sum().backward()
    (O.eval_sh_bases(deg, d2) * w).sum().backward()
    # the two polynomial forms agree on the unit sphere; their gradients agree along it (tangential part)
    tang = lambda v, n: v - (v * n).sum(-1, keepdim=True) * n
    assert (tang(d.grad, d.detach()) - tang(d2.grad, d.detach())).abs().max() < 1e-12
