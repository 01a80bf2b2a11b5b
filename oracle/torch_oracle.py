"""Float64 pure-PyTorch CPU restatement of the tile-based 3D-Gaussian-splatting rasteriser.

TEST INFRASTRUCTURE -- see oracle/__init__.py ("parity unpinned").

What it restates (the reference only *calls* this arithmetic, it does not contain it):
  * call contract                  /root/reference/utils/gsplat_utils/gsplat_trainer.py:446-497
                                   (`Runner.rasterize_splats` -> `gsplat.rendering.rasterization`)
  * published algorithm            Kerbl et al. 2023 (3DGS), Zwicker et al. (EWA splatting),
                                   gsplat maths supplement arXiv:2312.02121, gsplat paper arXiv:2409.06765
  * constants                      SURVEY.md Appendix B ([upstream-memory] of gsplat ~v1.4)

Every backward pass comes from torch.autograd over this forward code -- there are no
hand-written gradients here, so the oracle's gradients are correct whenever its forward is.
All discrete decisions (culling, radius, tile lists, alpha threshold, early stop) are
non-differentiable constants exactly as in the published algorithm.

Dtype: float64 by default (parity oracle).  `dtype=torch.float32` gives the timed CPU baseline.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
from torch import Tensor

# ----------------------------------------------------------------------------------------
# constants  (SH DC constant also at /root/reference/utils/gsplat_utils/utils.py:148-150)
# ----------------------------------------------------------------------------------------
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)
SH_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892,
         0.10578554691520431, -0.6690465435572892, 0.47308734787878004, -1.7701307697799304,
         0.6258357354491761)

ALPHA_MAX = 0.999          # alpha cap
ALPHA_MIN = 1.0 / 255.0    # contribution threshold
T_STOP = 1e-4              # transmittance early stop
FISHEYE_EPS = 1e-7

CAMERA_MODELS = ("pinhole", "ortho", "fisheye", "spherical")


# ----------------------------------------------------------------------------------------
# K1 projection
# ----------------------------------------------------------------------------------------
def quat_to_rotmat(quats: Tensor) -> Tensor:
    """(w,x,y,z) quaternion (normalised here) -> rotation matrix.  [...,4] -> [...,3,3]"""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    ], dim=-1)
    return R.reshape(quats.shape[:-1] + (3, 3))


def quat_scale_to_covar(quats: Tensor, scales: Tensor) -> Tensor:
    """Sigma = R diag(s^2) R^T.  [N,4],[N,3] -> [N,3,3]"""
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def _persp_proj(mc: Tensor, cc: Tensor, fx, fy, cx, cy, W: int, H: int):
    x, y, z = mc.unbind(-1)
    tan_fovx = 0.5 * W / fx
    tan_fovy = 0.5 * H / fy
    lim_x_pos = (W - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (H - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    rz = 1.0 / z
    rz2 = rz * rz
    tx = z * torch.minimum(lim_x_pos, torch.maximum(-lim_x_neg, x * rz))
    ty = z * torch.minimum(lim_y_pos, torch.maximum(-lim_y_neg, y * rz))
    O = torch.zeros_like(z)
    J = torch.stack([fx * rz, O, -fx * tx * rz2,
                     O, fy * rz, -fy * ty * rz2], dim=-1).reshape(z.shape + (2, 3))
    cov2d = J @ cc @ J.transpose(-1, -2)
    mean2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)
    return mean2d, cov2d


def _fisheye_proj(mc: Tensor, cc: Tensor, fx, fy, cx, cy, W: int, H: int):
    """Equidistant fisheye: r_img = f * theta, theta = atan2(|xy|, z)."""
    x, y, z = mc.unbind(-1)
    eps = FISHEYE_EPS
    xy_len = torch.sqrt(x * x + y * y) + eps
    theta = torch.atan2(xy_len, z + eps)
    mean2d = torch.stack([x * fx * theta / xy_len + cx, y * fy * theta / xy_len + cy], dim=-1)
    x2 = x * x + eps
    y2 = y * y
    xy = x * y
    x2y2 = x2 + y2
    x2y2z2_inv = 1.0 / (x2y2 + z * z)
    b = torch.atan2(xy_len, z) / xy_len / x2y2
    a = z * x2y2z2_inv / x2y2
    J = torch.stack([fx * (x2 * a + y2 * b), fx * xy * (a - b), -fx * x * x2y2z2_inv,
                     fy * xy * (a - b), fy * (y2 * a + x2 * b), -fy * y * x2y2z2_inv],
                    dim=-1).reshape(z.shape + (2, 3))
    # the Jacobian is a constant of the projection (EWA local affine approximation): like the
    # published implementation its entries ARE differentiated w.r.t. the mean.
    cov2d = J @ cc @ J.transpose(-1, -2)
    return mean2d, cov2d


def _ortho_proj(mc: Tensor, cc: Tensor, fx, fy, cx, cy, W: int, H: int):
    x, y, z = mc.unbind(-1)
    O = torch.zeros_like(z)
    fxb = fx + O
    fyb = fy + O
    J = torch.stack([fxb, O, O, O, fyb, O], dim=-1).reshape(z.shape + (2, 3))
    cov2d = J @ cc @ J.transpose(-1, -2)
    mean2d = torch.stack([fx * x + cx, fy * y + cy], dim=-1)
    return mean2d, cov2d


def _spherical_proj(mc: Tensor, cc: Tensor, fx, fy, cx, cy, W: int, H: int):
    """360-degree equirectangular camera (the reference's `spherical` / `equirectangular` data sets,
    utils/datasets/opensfm.py:176-193, 430-436).  The fork's kernel for it is absent from the reference tree, so this is
    the BUILD'S definition (parity unpinned): lon = atan2(x, z), lat = atan2(y, |xz|), u = W (lon/2pi + 1/2),
    v = H (lat/pi + 1/2); K is not used; EWA covariance with J = d(u,v)/d(x,y,z) written out (autograd differentiates
    its entries w.r.t. the mean, as for the other models)."""
    x, y, z = mc.unbind(-1)
    fxs, fys = W / (2.0 * math.pi), H / math.pi
    p2 = x * x + z * z + 1e-12
    p = torch.sqrt(p2)
    r2 = p2 + y * y
    mean2d = torch.stack([fxs * torch.atan2(x, z) + 0.5 * W, fys * torch.atan2(y, p) + 0.5 * H], dim=-1)
    k = fys / (r2 * p)
    O = torch.zeros_like(z)
    J = torch.stack([fxs * z / p2, O, -fxs * x / p2,
                     -k * x * y, fys * p / r2, -k * z * y], dim=-1).reshape(z.shape + (2, 3))
    cov2d = J @ cc @ J.transpose(-1, -2)
    return mean2d, cov2d


_PROJ = {"pinhole": _persp_proj, "fisheye": _fisheye_proj, "ortho": _ortho_proj, "spherical": _spherical_proj}


def fully_fused_projection(
    means: Tensor, covars: Optional[Tensor], quats: Optional[Tensor], scales: Optional[Tensor],
    viewmats: Tensor, Ks: Tensor, width: int, height: int,
    eps2d: float = 0.3, near_plane: float = 0.01, far_plane: float = 1e10,
    radius_clip: float = 0.0, calc_compensations: bool = False, camera_model: str = "pinhole",
    dtype: torch.dtype = torch.float64,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    """3D -> 2D EWA projection.

    Returns radii[C,N] int32, means2d[C,N,2], depths[C,N], conics[C,N,3], compensations[C,N]|None.
    Culled Gaussians have radii 0 and all-zero outputs (and receive zero gradient).
    """
    assert camera_model in CAMERA_MODELS, camera_model
    C, N = viewmats.shape[0], means.shape[0]
    means = means.to(dtype)
    viewmats = viewmats.to(dtype)
    Ks = Ks.to(dtype)
    if covars is None:
        covars = quat_scale_to_covar(quats.to(dtype), scales.to(dtype))
    else:
        covars = covars.to(dtype)
    Rm = viewmats[:, :3, :3]                                   # [C,3,3]
    tv = viewmats[:, :3, 3]                                    # [C,3]
    mc_all = torch.einsum("cij,nj->cni", Rm, means) + tv[:, None, :]      # [C,N,3]
    # what near / far apply to and what the tile sort orders by: camera-space z; the range for the spherical model
    depth_all = mc_all.norm(dim=-1) if camera_model == "spherical" else mc_all[..., 2]
    z_all = depth_all.detach()
    ok = (z_all >= near_plane) & (z_all <= far_plane)          # [C,N]
    ci, ni = torch.where(ok)
    mc = mc_all[ci, ni]                                        # [M,3]
    cc = Rm[ci] @ covars[ni] @ Rm[ci].transpose(-1, -2)        # [M,3,3]
    fx, fy, cx, cy = Ks[ci, 0, 0], Ks[ci, 1, 1], Ks[ci, 0, 2], Ks[ci, 1, 2]
    mean2d, cov2d = _PROJ[camera_model](mc, cc, fx, fy, cx, cy, width, height)

    a0, b0, c0, d0 = cov2d[:, 0, 0], cov2d[:, 0, 1], cov2d[:, 1, 0], cov2d[:, 1, 1]
    det_orig = a0 * d0 - b0 * c0
    a1 = a0 + eps2d
    d1 = d0 + eps2d
    det = a1 * d1 - b0 * c0
    good = det.detach() > 0
    det_s = torch.where(good, det, torch.ones_like(det))
    comp = torch.sqrt(torch.clamp(det_orig / det_s, min=0.0))
    conic = torch.stack([d1 / det_s, -b0 / det_s, a1 / det_s], dim=-1)
    bb = 0.5 * (a1 + d1)
    v1 = bb + torch.sqrt(torch.clamp(bb * bb - det_s, min=0.01))
    radius = torch.ceil(3.0 * torch.sqrt(v1)).detach()
    good = good & (radius > radius_clip)
    mx, my = mean2d[:, 0].detach(), mean2d[:, 1].detach()
    good = good & ~((mx + radius <= 0) | (mx - radius >= width) | (my + radius <= 0) | (my - radius >= height))

    gi = torch.where(good)[0]
    ci, ni = ci[gi], ni[gi]
    radii = torch.zeros(C, N, dtype=torch.int32)
    radii[ci, ni] = radius[gi].to(torch.int32)
    means2d = torch.zeros(C, N, 2, dtype=dtype).index_put((ci, ni), mean2d[gi])
    depths = torch.zeros(C, N, dtype=dtype).index_put((ci, ni), depth_all[ci, ni])
    conics = torch.zeros(C, N, 3, dtype=dtype).index_put((ci, ni), conic[gi])
    comps = None
    if calc_compensations:
        comps = torch.zeros(C, N, dtype=dtype).index_put((ci, ni), comp[gi])
    return radii, means2d, depths, conics, comps


# ----------------------------------------------------------------------------------------
# K4 spherical harmonics
# ----------------------------------------------------------------------------------------
def eval_sh_bases(degree: int, dirs: Tensor) -> Tensor:
    """Real SH basis with the 3DGS sign convention, for *unit* dirs.  [...,3] -> [...,(degree+1)^2]"""
    x, y, z = dirs.unbind(-1)
    out = [torch.full_like(x, SH_C0)]
    if degree >= 1:
        out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz,
                SH_C2[4] * (xx - yy)]
    if degree >= 3:
        out += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
                SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    if degree >= 4:
        out += [SH_C4[0] * xy * (xx - yy), SH_C4[1] * yz * (3 * xx - yy), SH_C4[2] * xy * (7 * zz - 1),
                SH_C4[3] * yz * (7 * zz - 3), SH_C4[4] * (zz * (35 * zz - 30) + 3),
                SH_C4[5] * xz * (7 * zz - 3), SH_C4[6] * (xx - yy) * (7 * zz - 1),
                SH_C4[7] * xz * (xx - 3 * yy), SH_C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(out, dim=-1)


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor,
                        masks: Optional[Tensor] = None, dtype: torch.dtype = torch.float64) -> Tensor:
    """colour = sum_k Y_k(dir/|dir|) * coeff_k for k < (degrees_to_use+1)^2.

    dirs [...,3] (normalised here), coeffs [...,K,3], masks [...] bool -> [...,3].
    Bands above `degrees_to_use` contribute nothing and receive zero gradient; masked-out
    entries are zero.
    """
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    dirs = dirs.to(dtype)
    coeffs = coeffs.to(dtype)
    if masks is not None:
        safe = torch.where(masks[..., None], dirs, torch.ones_like(dirs))
    else:
        safe = dirs
    d = safe / safe.norm(dim=-1, keepdim=True)
    nb = (degrees_to_use + 1) ** 2
    bases = eval_sh_bases(degrees_to_use, d)                       # [...,nb]
    col = (bases[..., None] * coeffs[..., :nb, :]).sum(dim=-2)
    if masks is not None:
        col = torch.where(masks[..., None], col, torch.zeros_like(col))
    return col


# ----------------------------------------------------------------------------------------
# K6-K8 tile binning, depth sort, offsets
# ----------------------------------------------------------------------------------------
def tile_bits(n_tiles: int) -> int:
    return int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 0


def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int,
                tile_width: int, tile_height: int, sort: bool = True, periodic: bool = False
                ) -> Tuple[Tensor, Tensor, Tensor]:
    """AABB tile overlap + 64-bit keys (cam | tile | fp32 depth bits) + stable sort.

    The AABB arithmetic is done in float32 (as the published kernel does) so that tile
    membership is bit-identical with a float32 device implementation fed the same float32
    means2d / radii.  Returns tiles_per_gauss[C,N] i32, isect_ids[I] i64, flatten_ids[I] i32.

    periodic (BUILD-DEFINED, like the `spherical` model itself -- the fork that defines it is absent, SURVEY.md
    section 8c): the image is periodic in x with period tile_width * tile_size.  The column range is then not clamped
    to the image but taken modulo tile_width (at most one full turn: a box wider than the image covers every column
    once: the tile_width columns whose centres lie within half an image of its centre), so a footprint that crosses the +-pi seam of an
    equirectangular panorama reaches the tiles on the other side; `rasterize_to_pixels(periodic=True)` evaluates it there
    at its nearest copy.
    """
    C, N = radii.shape
    m = means2d.detach().to(torch.float32)
    r = radii.to(torch.float32)
    ts = torch.tensor(float(tile_size), dtype=torch.float32)
    tile_r = r / ts
    tx = m[..., 0] / ts
    ty = m[..., 1] / ts
    if periodic:
        x0 = torch.floor(tx - tile_r).clamp(min=-tile_width).to(torch.int64)
        x1 = torch.ceil(tx + tile_r).clamp(max=2 * tile_width).to(torch.int64)
        full = (x1 - x0) > tile_width                 # wider than the image: the tile_width columns centred on the splat
        xc = torch.ceil(tx - 0.5 * tile_width - 0.5).to(torch.int64)     # column centres within half an image of the splat
        x0 = torch.where(full, xc, x0)
        x1 = torch.where(full, xc + tile_width, x1)
    else:
        x0 = torch.floor(tx - tile_r).clamp(0, tile_width).to(torch.int64)
        x1 = torch.ceil(tx + tile_r).clamp(0, tile_width).to(torch.int64)
    y0 = torch.floor(ty - tile_r).clamp(0, tile_height).to(torch.int64)
    y1 = torch.ceil(ty + tile_r).clamp(0, tile_height).to(torch.int64)
    vis = radii > 0
    nx = torch.where(vis, x1 - x0, torch.zeros_like(x0))
    ny = torch.where(vis, y1 - y0, torch.zeros_like(y0))
    tpg = (nx * ny)
    tiles_per_gauss = tpg.to(torch.int32)
    flat_tpg = tpg.reshape(-1)
    total = int(flat_tpg.sum())
    n_tiles = tile_width * tile_height
    tb = tile_bits(n_tiles)
    if total == 0:
        return tiles_per_gauss, torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32)
    gid = torch.repeat_interleave(torch.arange(C * N), flat_tpg)            # flatten id per isect
    start = torch.cumsum(flat_tpg, 0) - flat_tpg
    local = torch.arange(total) - start[gid]
    nxg = nx.reshape(-1)[gid]
    iy = y0.reshape(-1)[gid] + local // nxg                                  # row-major emission
    ix = x0.reshape(-1)[gid] + local % nxg
    if periodic:
        ix = ix % tile_width                                                 # Python modulo: -1 -> tile_width - 1
    tile_id = iy * tile_width + ix
    cam = gid // N
    dbits = depths.detach().to(torch.float32).reshape(-1)[gid].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    isect_ids = (cam << (32 + tb)) | (tile_id << 32) | dbits
    flatten_ids = gid.to(torch.int32)
    if sort:
        order = torch.sort(isect_ids, stable=True).indices
        isect_ids = isect_ids[order]
        flatten_ids = flatten_ids[order]
    return tiles_per_gauss, isect_ids, flatten_ids


def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """offsets[c,ty,tx] = first index in the sorted list whose (cam,tile) >= (c,ty,tx)."""
    n_tiles = tile_width * tile_height
    tb = tile_bits(n_tiles)
    key = isect_ids >> 32
    cam = key >> tb
    tile = key & ((1 << tb) - 1)
    lin = cam * n_tiles + tile
    q = torch.arange(n_cameras * n_tiles, dtype=torch.int64)
    off = torch.searchsorted(lin.contiguous(), q, right=False)
    return off.to(torch.int32).reshape(n_cameras, tile_height, tile_width)


# ----------------------------------------------------------------------------------------
# K9 rasterise
# ----------------------------------------------------------------------------------------
def rasterize_to_pixels(
    means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor,
    width: int, height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
    backgrounds: Optional[Tensor] = None, absgrad_probe: Optional[List] = None,
    dtype: torch.dtype = torch.float64, return_last_ids: bool = False, periodic: bool = False,
):
    """Front-to-back alpha compositing.
    periodic: see `isect_tiles` -- inside tile (ty, tx) a Gaussian at x is evaluated at the copy
    x - width * round((x - tile centre x) / width) (round half to even), width % tile_size == 0 required.

    means2d[C,N,2] conics[C,N,3] colors[C,N,D] opacities[C,N] -> colors[C,H,W,D], alphas[C,H,W,1].
    If `absgrad_probe` is a list, per-(Gaussian,pixel) zero offsets are spliced into the
    pixel-to-mean delta and recorded there; after backward, `collect_absgrad` sums the
    absolute per-pixel gradients (the `absgrad` statistic of the densification strategy).
    """
    C, N = opacities.shape
    D = colors.shape[-1]
    m2 = means2d.to(dtype).reshape(C * N, 2)
    cn = conics.to(dtype).reshape(C * N, 3)
    col = colors.to(dtype).reshape(C * N, D)
    op = opacities.to(dtype).reshape(C * N)
    th, tw = isect_offsets.shape[1:]
    off = isect_offsets.reshape(-1).tolist() + [int(flatten_ids.numel())]
    fid = flatten_ids.to(torch.int64)
    P = height * width
    out_pix: List[Tensor] = []
    out_col: List[Tensor] = []
    out_T: List[Tensor] = []
    last_ids = torch.zeros(C * P, dtype=torch.int32)
    n_tiles = th * tw
    for c in range(C):
        for t in range(n_tiles):
            lo, hi = off[c * n_tiles + t], off[c * n_tiles + t + 1]
            if hi <= lo:
                continue
            ty, tx = divmod(t, tw)
            ii = torch.arange(ty * tile_size, min((ty + 1) * tile_size, height))
            jj = torch.arange(tx * tile_size, min((tx + 1) * tile_size, width))
            gi, gj = torch.meshgrid(ii, jj, indexing="ij")
            pix = (gi * width + gj).reshape(-1)                                   # [p]
            px = gj.reshape(-1).to(dtype) + 0.5
            py = gi.reshape(-1).to(dtype) + 0.5
            g = fid[lo:hi]                                                        # [L]
            mx = m2[g, 0]
            if periodic:
                mx = mx - width * torch.round((mx.detach() - (tx * tile_size + 0.5 * tile_size)) / width)
            dx = mx[:, None] - px[None, :]                                        # [L,p]
            dy = m2[g, 1][:, None] - py[None, :]
            if absgrad_probe is not None:
                e = torch.zeros(g.numel(), pix.numel(), 2, dtype=dtype, requires_grad=True)
                absgrad_probe.append((g, e))
                dx = dx + e[..., 0]
                dy = dy + e[..., 1]
            ca, cb, cc = cn[g, 0][:, None], cn[g, 1][:, None], cn[g, 2][:, None]
            sigma = 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
            alpha = torch.clamp(op[g][:, None] * torch.exp(-sigma), max=ALPHA_MAX)
            valid = (sigma.detach() >= 0) & (alpha.detach() >= ALPHA_MIN)
            a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
            Tincl = torch.cumprod(1.0 - a_eff, dim=0)                             # T after k
            incl = valid & (Tincl.detach() > T_STOP)
            # everything after the first stop is excluded (Tincl is non-increasing, so the
            # `> T_STOP` test already is monotone)
            a_use = torch.where(incl, alpha, torch.zeros_like(alpha))
            Tin = torch.cumprod(1.0 - a_use, dim=0)
            Tex = torch.cat([torch.ones_like(Tin[:1]), Tin[:-1]], dim=0)          # T before k
            wgt = a_use * Tex                                                     # [L,p]
            out_col.append(wgt.transpose(0, 1) @ col[g])                          # [p,D]
            out_T.append(Tin[-1])
            out_pix.append(pix + c * P)
            # last contributing index (absolute position in the sorted list)
            pos = torch.arange(lo, hi)[:, None].expand_as(incl)
            last = torch.where(incl, pos, torch.zeros_like(pos)).max(dim=0).values
            last_ids[pix + c * P] = last.to(torch.int32)
    acc = torch.zeros(C * P, D, dtype=dtype)
    Tfin = torch.ones(C * P, dtype=dtype)
    if out_pix:
        ids = torch.cat(out_pix)
        acc = acc.index_put((ids,), torch.cat(out_col))
        Tfin = Tfin.index_put((ids,), torch.cat(out_T))
    acc = acc.reshape(C, height, width, D)
    Tfin = Tfin.reshape(C, height, width, 1)
    if backgrounds is not None:
        acc = acc + Tfin * backgrounds.to(dtype)[:, None, None, :]
    alphas = 1.0 - Tfin
    if return_last_ids:
        return acc, alphas, last_ids.reshape(C, height, width)
    return acc, alphas


def collect_absgrad(absgrad_probe: List, n_flat: int, dtype=torch.float64) -> Tensor:
    """Sum over pixels of |d loss / d means2d| per flattened (camera, Gaussian).  -> [n_flat,2]"""
    out = torch.zeros(n_flat, 2, dtype=dtype)
    for g, e in absgrad_probe:
        if e.grad is not None:
            out.index_add_(0, g, e.grad.abs().sum(dim=1))
    return out


# ----------------------------------------------------------------------------------------
# the full path (what `Runner.rasterize_splats` reaches at gsplat_trainer.py:477-494)
# ----------------------------------------------------------------------------------------
def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor,
    viewmats: Tensor, Ks: Tensor, width: int, height: int,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
    eps2d: float = 0.3, sh_degree: Optional[int] = None, tile_size: int = 16,
    backgrounds: Optional[Tensor] = None, render_mode: str = "RGB",
    rasterize_mode: str = "classic", camera_model: str = "pinhole",
    covars: Optional[Tensor] = None, absgrad_probe: Optional[List] = None,
    dtype: torch.dtype = torch.float64, raster_fn=None, sort_depths: Optional[Tensor] = None,
):
    """means[N,3] quats[N,4] scales[N,3] opacities[N] colors[N,K,3]|[N,D]|[C,N,D] ->
    (render_colors[C,H,W,X], render_alphas[C,H,W,1], meta).  Activations (exp / sigmoid) are
    the caller's job, as at gsplat_trainer.py:456-459.

    sort_depths [C,N] (optional): float32 depths to build the tile-sort keys from INSTEAD of this function's own
    depths rounded to float32.  The sort key is a discrete input of the algorithm (fp32 depth bits, SURVEY.md B.1
    step 6): two overlapping Gaussians whose depths differ by less than one float32 ulp are ordered by whatever the
    float32 depth computation rounds to, and a float64 restatement rounds differently on a handful of pairs per
    100k Gaussians.  A parity test that hands the device's own float32 depths in here (after checking that they
    are within 2 ulp of the float64 values, `depth_key_report`) compares the arithmetic on identical blending
    orders; everything differentiable still uses this function's own float64 depths."""
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED"), render_mode
    assert rasterize_mode in ("classic", "antialiased"), rasterize_mode
    C, N = viewmats.shape[0], means.shape[0]
    radii, means2d, depths, conics, comps = fully_fused_projection(
        means, covars, quats, scales, viewmats, Ks, width, height, eps2d=eps2d,
        near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip,
        calc_compensations=(rasterize_mode == "antialiased"), camera_model=camera_model, dtype=dtype)
    opac = opacities.to(dtype)[None, :].expand(C, N)
    if comps is not None:
        opac = opac * comps
    if sh_degree is None:
        cols = colors.to(dtype)
        if cols.dim() == 2:
            cols = cols[None].expand(C, -1, -1)
    else:
        campos = torch.linalg.inv(viewmats.to(dtype))[:, :3, 3]
        dirs = means.to(dtype)[None, :, :] - campos[:, None, :]
        shs = colors.to(dtype)
        if shs.dim() == 3:
            shs = shs[None].expand(C, -1, -1, -1)
        cols = spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0, dtype=dtype)
        cols = torch.clamp_min(cols + 0.5, 0.0)
    bg = backgrounds
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if bg is not None:
            bg = torch.cat([bg.to(dtype), torch.zeros(C, 1, dtype=dtype)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if bg is not None:
            bg = torch.zeros(C, 1, dtype=dtype)
    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    key_depths = depths
    if sort_depths is not None:
        assert sort_depths.shape == depths.shape, (sort_depths.shape, depths.shape)
        key_depths = sort_depths.detach().to("cpu", torch.float32)
    # build-defined: a panorama is periodic in x when the tile grid lines up across the seam
    periodic = camera_model == "spherical" and width % tile_size == 0
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(
        means2d, radii, key_depths, tile_size, tile_width, tile_height, periodic=periodic)
    isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)
    meta = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opac,
                tile_width=tile_width, tile_height=tile_height, tiles_per_gauss=tiles_per_gauss,
                isect_ids=isect_ids, flatten_ids=flatten_ids, isect_offsets=isect_offsets,
                width=width, height=height, tile_size=tile_size, n_cameras=C, colors=cols)
    if raster_fn is None:
        render_colors, render_alphas = rasterize_to_pixels(
            means2d, conics, cols, opac, width, height, tile_size, isect_offsets, flatten_ids,
            backgrounds=bg, absgrad_probe=absgrad_probe, dtype=dtype, periodic=periodic)
    else:
        kw = {"periodic": True} if periodic else {}
        render_colors, render_alphas = raster_fn(
            means2d, conics, cols, opac, width, height, tile_size, isect_offsets, flatten_ids, bg, **kw)
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat([render_colors[..., :-1],
                                   render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
    return render_colors, render_alphas, meta


def depth_key_report(depths_oracle: Tensor, depths_device: Tensor, radii: Tensor) -> dict:
    """How far the device's float32 depths are from this oracle's (float64) depths, in float32 ulps, over the
    visible (camera, Gaussian) pairs; and how many ORDERED pairs of the per-camera depth ranking the two disagree on
    (counted over adjacent entries of the device's ranking -- a lower bound of the blending-order flips)."""
    vis = radii > 0
    d64 = depths_oracle.detach().double()[vis]
    d32 = depths_device.detach().cpu().float()[vis]
    ulp = torch.abs(torch.nextafter(d32, torch.full_like(d32, float("inf"))) - d32).double()
    err_ulp = (d32.double() - d64).abs() / ulp
    r32 = d64.float()
    flips = 0
    for c in range(radii.shape[0]):
        v = vis[c]
        a = depths_device[c].detach().cpu().float()[v]
        b = depths_oracle[c].detach().double()[v]
        order = torch.argsort(a, stable=True)
        flips += int((b[order][1:] < b[order][:-1]).sum())
    return {"max_ulp": float(err_ulp.max()) if err_ulp.numel() else 0.0,
            "keys_differ": int((r32.view(torch.int32) != d32.view(torch.int32)).sum()),
            "adjacent_rank_flips": flips, "visible": int(vis.sum())}
