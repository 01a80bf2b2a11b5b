"""Operator surface of the path -- same names, arguments and error behaviour as the gsplat
operators the reference reaches through `rasterization`:

    fully_fused_projection (legacy alias project_gaussians), spherical_harmonics, isect_tiles,
    isect_offset_encode, rasterize_to_pixels

Each is a thin torch.autograd.Function over the C ABI of libsplat_one_amd.so
(include/splat_one_amd.h).  PyTorch supplies device memory, the current HIP stream and the
autograd tape; all arithmetic runs in the hand-written gfx950 kernels.  There is no CPU path.

Reference call site: /root/reference/utils/gsplat_utils/gsplat_trainer.py:477-494.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch
from torch import Tensor

from . import _lib
from ._lib import call, ptr, stream

CAMERA_MODELS = {"pinhole": 0, "ortho": 1, "fisheye": 2, "spherical": 3}    # spherical: defined by this build, see splat_math.hpp
SUPPORTED_CHANNELS = (1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33)


def _f32(t: Tensor) -> Tensor:
    assert t.dtype == torch.float32, f"expected float32, got {t.dtype}"
    return t.contiguous()


def _camera_model_id(camera_model: str) -> int:
    assert camera_model in CAMERA_MODELS, (
        f"camera_model must be one of {list(CAMERA_MODELS)}, got {camera_model!r}")
    return CAMERA_MODELS[camera_model]


SO_CAM_PER_VIEW, SO_CAM_PER_VIEW_MAX = 0x40000000, 15


def camera_model_code(camera_model, n_views: int) -> int:
    """`camera_model` argument of the fused entry points: one name for all views, or one name per view
    (mixed perspective / fisheye batches; include/splat_one_amd.h SO_CAM_PER_VIEW)."""
    if isinstance(camera_model, str):
        return _camera_model_id(camera_model)
    models = list(camera_model)
    assert len(models) == n_views, f"{len(models)} camera models for {n_views} views"
    if len(set(models)) == 1:
        return _camera_model_id(models[0])
    assert n_views <= SO_CAM_PER_VIEW_MAX, f"per-view camera models: at most {SO_CAM_PER_VIEW_MAX} views per step"
    code = SO_CAM_PER_VIEW
    for c, m in enumerate(models):
        code |= _camera_model_id(m) << (2 * c)
    return code


# ---------------------------------------------------------------------------------------------
# inverse(camtoworlds) without torch.linalg.inv's device synchronisation (its singularity check reads back)
# ---------------------------------------------------------------------------------------------
class _CameraInverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, camtoworlds: Tensor) -> Tensor:
        c2w = _f32(camtoworlds)
        out = torch.empty_like(c2w)
        call("so_camera_inverse", c2w.shape[0], ptr(c2w), ptr(out), stream())
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, v_out: Tensor):
        (inv,) = ctx.saved_tensors                       # d(A^-1) = -A^-1 dA A^-1  ->  v_A = -A^-T v_out A^-T
        it = inv.transpose(-1, -2)
        return -(it @ v_out @ it)


def camera_inverse(camtoworlds: Tensor) -> Tensor:
    """viewmats[C,4,4] = inverse(camtoworlds[C,4,4]) (general 4x4, computed in double on the device; differentiable --
    pose optimisation reaches the camera deltas through it).  gsplat_trainer.py:483 uses torch.linalg.inv, which
    synchronises the device to report singular inputs."""
    assert camtoworlds.shape[-2:] == (4, 4) and camtoworlds.dim() == 3, camtoworlds.shape
    return _CameraInverse.apply(camtoworlds)


# ---------------------------------------------------------------------------------------------
# K1/K2 projection
# ---------------------------------------------------------------------------------------------
def _covars_to_6(covars: Tensor) -> Tensor:
    """[N,3,3] (symmetric) or [N,6] -> [N,6] (xx,xy,xz,yy,yz,zz)."""
    if covars.shape[-2:] == (3, 3):
        i = torch.tensor([0, 0, 0, 1, 1, 2], device=covars.device)
        j = torch.tensor([0, 1, 2, 1, 2, 2], device=covars.device)
        return covars[..., i, j]
    assert covars.shape[-1] == 6, covars.shape
    return covars


class _FullyFusedProjection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, covars6, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, calc_compensations, camera_model):
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        radii = torch.empty(C, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(C, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(C, N, dtype=torch.float32, device=dev)
        conics = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        comps = torch.empty(C, N, dtype=torch.float32, device=dev) if calc_compensations else None
        call("so_projection_fwd", C, N, ptr(means), ptr(covars6), ptr(quats), ptr(scales), ptr(viewmats),
             ptr(Ks), width, height, eps2d, near_plane, far_plane, radius_clip, camera_model, ptr(radii),
             ptr(means2d), ptr(depths), ptr(conics), ptr(comps), stream())
        ctx.save_for_backward(means, covars6, quats, scales, viewmats, Ks, radii, conics, comps)
        ctx.cfg = (width, height, eps2d, camera_model)
        ctx.mark_non_differentiable(radii)
        if comps is None:
            return radii, means2d, depths, conics
        return radii, means2d, depths, conics, comps

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps=None):
        means, covars6, quats, scales, viewmats, Ks, radii, conics, comps = ctx.saved_tensors
        width, height, eps2d, camera_model = ctx.cfg
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        z = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else t.contiguous()
        v_means2d = z(v_means2d, (C, N, 2))
        v_depths = z(v_depths, (C, N))
        v_conics = z(v_conics, (C, N, 3))
        v_comps = None if comps is None else z(v_comps, (C, N))
        v_means = torch.empty_like(means)
        v_covars6 = torch.empty_like(covars6) if covars6 is not None else None
        v_quats = torch.empty_like(quats) if covars6 is None else None
        v_scales = torch.empty_like(scales) if covars6 is None else None
        v_viewmats = torch.zeros_like(viewmats) if ctx.needs_input_grad[4] else None
        call("so_projection_bwd", C, N, ptr(means), ptr(covars6), ptr(quats), ptr(scales), ptr(viewmats),
             ptr(Ks), width, height, eps2d, camera_model, ptr(radii), ptr(v_means2d), ptr(v_depths),
             ptr(v_conics), ptr(v_comps), ptr(v_means), ptr(v_covars6), ptr(v_quats), ptr(v_scales),
             ptr(v_viewmats), stream())
        return (v_means, v_covars6, v_quats, v_scales, v_viewmats) + (None,) * 9


def _sparse_rows(dense: Optional[Tensor], rows: Tensor) -> Optional[Tensor]:
    """Row-sparse COO view of a per-Gaussian gradient: the rows `rows` (unique, sorted) of `dense`."""
    if dense is None:
        return None
    return torch.sparse_coo_tensor(rows[None], dense[rows], size=dense.shape, is_coalesced=True)


class _PackedProjection(torch.autograd.Function):
    """Packed layout of gsplat's `fully_fused_projection(packed=True)`: one row per (camera, Gaussian) pair with a
    positive radius, in ascending flattened index (camera-major).  Two passes of `so_projection_packed` over the raw
    inputs -- count + scan, then write -- with ONE host read (nnz, to size the outputs) and no [C,N] array; the backward
    (`so_projection_bwd_packed`) works on the nnz rows too."""

    @staticmethod
    def forward(ctx, means, covars6, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, calc_compensations, camera_model, sparse_grad):
        from . import _lib
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        nblk = int(_lib.load().so_projection_packed_blocks(C, N))
        counts = torch.empty(max(nblk, 1), dtype=torch.int32, device=dev)
        offsets = torch.empty(max(nblk, 1), dtype=torch.int64, device=dev)
        masks = torch.empty(16 * max(nblk, 1), dtype=torch.int64, device=dev)      # one bit per (camera, Gaussian) pair
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        common = (C, N, ptr(means), ptr(covars6), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), width, height, eps2d,
                  near_plane, far_plane, radius_clip, camera_model)
        call("so_projection_packed", *common, ptr(counts), ptr(masks), ptr(offsets), ptr(total), 0, 0, 0, 0, 0, 0, 0, stream())
        nnz = int(total.item())                                   # the one host read: sizes of the returned tensors
        camera_ids = torch.empty(nnz, dtype=torch.int64, device=dev)
        gaussian_ids = torch.empty(nnz, dtype=torch.int64, device=dev)
        radii_p = torch.empty(nnz, dtype=torch.int32, device=dev)
        means2d_p = torch.empty(nnz, 2, dtype=torch.float32, device=dev)
        depths_p = torch.empty(nnz, dtype=torch.float32, device=dev)
        conics_p = torch.empty(nnz, 3, dtype=torch.float32, device=dev)
        comps_p = torch.empty(nnz, dtype=torch.float32, device=dev) if calc_compensations else None
        if nnz:
            call("so_projection_packed", *common, 0, ptr(masks), ptr(offsets), 0, ptr(camera_ids), ptr(gaussian_ids), ptr(radii_p),
                 ptr(means2d_p), ptr(depths_p), ptr(conics_p), ptr(comps_p), stream())
        ctx.save_for_backward(means, covars6, quats, scales, viewmats, Ks, camera_ids, gaussian_ids)
        ctx.cfg = (width, height, eps2d, camera_model, sparse_grad, comps_p is not None)
        ctx.mark_non_differentiable(camera_ids, gaussian_ids, radii_p)
        if comps_p is None:
            return camera_ids, gaussian_ids, radii_p, means2d_p, depths_p, conics_p
        return camera_ids, gaussian_ids, radii_p, means2d_p, depths_p, conics_p, comps_p

    @staticmethod
    def backward(ctx, _v_cid, _v_gid, _v_radii, v_means2d, v_depths, v_conics, v_comps=None):
        means, covars6, quats, scales, viewmats, Ks, camera_ids, gaussian_ids = ctx.saved_tensors
        width, height, eps2d, camera_model, sparse_grad, has_comps = ctx.cfg
        C, N = viewmats.shape[0], means.shape[0]
        nnz = camera_ids.shape[0]
        dev = means.device
        z = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else _f32(t)
        v_means2d, v_depths, v_conics = z(v_means2d, (nnz, 2)), z(v_depths, (nnz,)), z(v_conics, (nnz, 3))
        v_comps = z(v_comps, (nnz,)) if has_comps else None
        v_means = torch.zeros_like(means)
        v_covars6 = torch.zeros_like(covars6) if covars6 is not None else None
        v_quats = torch.zeros_like(quats) if covars6 is None else None
        v_scales = torch.zeros_like(scales) if covars6 is None else None
        v_viewmats = torch.zeros_like(viewmats) if ctx.needs_input_grad[4] else None
        call("so_projection_bwd_packed", C, N, nnz, ptr(means), ptr(covars6), ptr(quats), ptr(scales), ptr(viewmats),
             ptr(Ks), width, height, eps2d, camera_model, ptr(camera_ids), ptr(gaussian_ids), ptr(v_means2d), ptr(v_depths),
             ptr(v_conics), ptr(v_comps), ptr(v_means), ptr(v_covars6), ptr(v_quats), ptr(v_scales), ptr(v_viewmats),
             stream())
        if sparse_grad:      # rows of Gaussians no camera sees are exactly zero: hand back only the visible rows
            rows = gaussian_ids if C == 1 else torch.unique(gaussian_ids)
            v_means, v_covars6 = _sparse_rows(v_means, rows), _sparse_rows(v_covars6, rows)
            v_quats, v_scales = _sparse_rows(v_quats, rows), _sparse_rows(v_scales, rows)
        return (v_means, v_covars6, v_quats, v_scales, v_viewmats) + (None,) * 10


def fully_fused_projection(
    means: Tensor, covars: Optional[Tensor], quats: Optional[Tensor], scales: Optional[Tensor],
    viewmats: Tensor, Ks: Tensor, width: int, height: int, eps2d: float = 0.3,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, packed: bool = False,
    sparse_grad: bool = False, calc_compensations: bool = False, camera_model: str = "pinhole",
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    """means[N,3], covars[N,6]|[N,3,3]|None, quats[N,4], scales[N,3], viewmats[C,4,4], Ks[C,3,3] ->
    (radii[C,N] i32, means2d[C,N,2], depths[C,N], conics[C,N,3], compensations[C,N]|None);
    packed=True (gsplat_trainer.py:487 `packed=self.cfg.packed`) -> (camera_ids[nnz] i64, gaussian_ids[nnz] i64,
    radii[nnz], means2d[nnz,2], depths[nnz], conics[nnz,3], compensations[nnz]|None) for the pairs with radius > 0,
    camera-major; sparse_grad=True (packed only) returns the gradients of means / quats / scales (covars) as
    row-sparse COO tensors over the visible Gaussians."""
    C, N = viewmats.shape[0], means.shape[0]
    assert means.shape == (N, 3), means.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert packed or not sparse_grad, "sparse_grad requires packed=True"
    if covars is not None:
        covars6 = _f32(_covars_to_6(covars))
        assert covars6.shape == (N, 6), covars6.shape
        quats = scales = None
    else:
        covars6 = None
        assert quats is not None and scales is not None, "covars or quats/scales is required"
        assert quats.shape == (N, 4), quats.shape
        assert scales.shape == (N, 3), scales.shape
        quats, scales = _f32(quats), _f32(scales)
    if packed:
        out = _PackedProjection.apply(
            _f32(means), covars6, quats, scales, _f32(viewmats), _f32(Ks), int(width), int(height),
            float(eps2d), float(near_plane), float(far_plane), float(radius_clip), bool(calc_compensations),
            _camera_model_id(camera_model), bool(sparse_grad))
        return out + (None,) if len(out) == 6 else out
    out = _FullyFusedProjection.apply(
        _f32(means), covars6, quats, scales, _f32(viewmats), _f32(Ks), int(width), int(height),
        float(eps2d), float(near_plane), float(far_plane), float(radius_clip), bool(calc_compensations),
        _camera_model_id(camera_model))
    if len(out) == 4:
        return out + (None,)
    return out


project_gaussians = fully_fused_projection  # legacy gsplat name used by north_star


# ---------------------------------------------------------------------------------------------
# K4/K5 spherical harmonics
# ---------------------------------------------------------------------------------------------
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs, masks, per_camera, C, N):
        K = coeffs.shape[-2]
        colors = torch.empty(dirs.shape, dtype=torch.float32, device=dirs.device)
        call("so_sh_fwd", C, N, K, degree, ptr(dirs), ptr(coeffs), per_camera, ptr(masks), ptr(colors), stream())
        ctx.save_for_backward(dirs, coeffs, masks)
        ctx.cfg = (degree, per_camera, C, N, K)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        dirs, coeffs, masks = ctx.saved_tensors
        degree, per_camera, C, N, K = ctx.cfg
        v_colors = v_colors.contiguous()
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if ctx.needs_input_grad[1] else None
        call("so_sh_bwd", C, N, K, degree, ptr(dirs), ptr(coeffs), per_camera, ptr(masks), ptr(v_colors),
             ptr(v_coeffs), ptr(v_dirs), stream())
        return None, v_dirs, v_coeffs, None, None, None, None


class _ShViewColors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, means, campos, coeffs, radii):
        C, N, K = campos.shape[0], means.shape[0], coeffs.shape[-2]
        colors = torch.empty(C, N, 3, dtype=torch.float32, device=means.device)
        call("so_sh_view_colors_fwd", C, N, K, degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii), ptr(colors), stream())
        ctx.save_for_backward(means, campos, coeffs, radii, colors)
        ctx.degree = degree
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        means, campos, coeffs, radii, colors = ctx.saved_tensors
        C, N, K = campos.shape[0], means.shape[0], coeffs.shape[-2]
        v_coeffs = torch.empty_like(coeffs)
        v_means = torch.empty_like(means)
        call("so_sh_view_colors_bwd", C, N, K, ctx.degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii), ptr(colors),
             ptr(v_colors.contiguous()), ptr(v_coeffs), ptr(v_means), stream())
        return None, v_means, None, v_coeffs, None


def sh_view_colors(degrees_to_use: int, means: Tensor, campos: Tensor, coeffs: Tensor, radii: Optional[Tensor]) -> Tensor:
    """The colour stage of `rasterization` fused: clamp_min(spherical_harmonics(degree, means[None] - campos[:, None], coeffs,
    masks=radii > 0) + 0.5, 0) -> colors[C,N,3] in ONE launch each way (six torch-level operations otherwise).  means[N,3],
    campos[C,3] (no gradient flows to it: callers that optimise poses use the unfused operators), coeffs[N,K,3] shared by
    the cameras, radii[C,N] int32 or None."""
    assert means.dim() == 2 and campos.dim() == 2 and coeffs.dim() == 3 and coeffs.shape[0] == means.shape[0], (means.shape, coeffs.shape)
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert radii is None or (radii.shape == (campos.shape[0], means.shape[0]) and radii.dtype == torch.int32), radii
    return _ShViewColors.apply(int(degrees_to_use), _f32(means), _f32(campos.detach()), _f32(coeffs),
                               None if radii is None else radii.contiguous())


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """dirs[...,3], coeffs[...,K,3], masks[...] -> colors[...,3].

    Leading dims are either equal, or coeffs is an expanded (stride-0) view over the first dim
    of dirs -- the `shs = colors.expand(C,...)` pattern of `rasterization` -- which is evaluated
    without materialising C copies."""
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3 and coeffs.shape[-1] == 3, (dirs.shape, coeffs.shape)
    if masks is not None:
        assert masks.shape == dirs.shape[:-1], masks.shape
        masks = masks.to(torch.uint8).contiguous() if masks.dtype != torch.uint8 else masks.contiguous()
    lead = dirs.shape[:-1]
    shared = coeffs.dim() >= 4 and coeffs.stride(0) == 0 and dirs.dim() >= 3
    if shared:
        C = lead[0]
        N = int(torch.Size(lead[1:]).numel())
        c = _f32(coeffs[0]).reshape(N, coeffs.shape[-2], 3)
        per_camera = 0
    else:
        C, N = 1, int(torch.Size(lead).numel())
        c = _f32(coeffs).reshape(N, coeffs.shape[-2], 3)
        per_camera = 0
    d = _f32(dirs).reshape(C, N, 3)
    m = None if masks is None else masks.reshape(C, N)
    out = _SphericalHarmonics.apply(int(degrees_to_use), d, c, m, per_camera, C, N)
    return out.reshape(lead + (3,))


def _eval_sh_bases_fast(basis_dim: int, dirs: Tensor) -> Tensor:
    """Real SH basis values Y_k(dir) for the first `basis_dim` in {1, 4, 9, 16, 25} functions, 3DGS sign convention,
    for UNIT `dirs[..., 3]` -> [..., basis_dim].  Drop-in for `gsplat.cuda._torch_impl._eval_sh_bases_fast`, which the
    reference's appearance module imports (/root/reference/utils/gsplat_utils/utils.py:91, 107).  Elementwise torch
    arithmetic on whatever device `dirs` lives on, differentiable by autograd like the function it replaces (the colour
    path proper evaluates the basis inside so_sh_fwd / so_preprocess_fwd)."""
    assert basis_dim in (1, 4, 9, 16, 25), f"basis_dim {basis_dim}: (degree + 1)^2 for degree 0..4 expected"
    assert dirs.shape[-1] == 3, dirs.shape
    x, y, z = dirs.unbind(-1)
    out = [torch.full_like(x, 0.2820947917738781)]
    if basis_dim > 1:
        c1 = 0.48860251190292
        out += [-c1 * y, c1 * z, -c1 * x]
    if basis_dim > 4:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out += [1.092548430592079 * xy, -1.092548430592079 * yz, 0.9461746957575601 * zz - 0.3153915652525201,
                -1.092548430592079 * xz, 0.5462742152960395 * (xx - yy)]
    if basis_dim > 9:
        out += [-0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * xy * z,
                -0.4570457994644658 * y * (5 * zz - 1), 0.3731763325901154 * z * (5 * zz - 3),
                -0.4570457994644658 * x * (5 * zz - 1), 1.445305721320277 * z * (xx - yy),
                -0.5900435899266435 * x * (xx - 3 * yy)]
    if basis_dim > 16:
        out += [2.5033429417967046 * xy * (xx - yy), -1.7701307697799304 * yz * (3 * xx - yy),
                0.9461746957575601 * xy * (7 * zz - 1), -0.6690465435572892 * yz * (7 * zz - 3),
                0.10578554691520431 * (zz * (35 * zz - 30) + 3), -0.6690465435572892 * xz * (7 * zz - 3),
                0.47308734787878004 * (xx - yy) * (7 * zz - 1), -1.7701307697799304 * xz * (xx - 3 * yy),
                0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(out, dim=-1)


# ---------------------------------------------------------------------------------------------
# K6-K8 tile binning / sort / offsets  (non-differentiable)
# ---------------------------------------------------------------------------------------------
SO_TILE_WRAP_ALL = 1 << 24          # include/splat_one_amd.h: every camera's image is periodic in x


def _periodic_tile_size(tile_size: int, periodic: bool) -> int:
    """The C ABI carries the periodic-image flag in the high bits of its `tile_size` argument."""
    return int(tile_size) | (SO_TILE_WRAP_ALL if periodic else 0)


@torch.no_grad()
def isect_tiles(
    means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int, tile_height: int,
    sort: bool = True, packed: bool = False, n_cameras: Optional[int] = None,
    camera_ids: Optional[Tensor] = None, gaussian_ids: Optional[Tensor] = None,
    return_offsets: bool = False, periodic: bool = False,
    conics: Optional[Tensor] = None, opacities: Optional[Tensor] = None,
):
    """means2d[C,N,2], radii[C,N] i32, depths[C,N] -> (tiles_per_gauss[C,N] i32, isect_ids[I] i64,
    flatten_ids[I] i32).  Exact-size outputs need I on the host: this entry point performs ONE
    device->host read (the sync-free path with preallocated capacity is `isect_tiles_static`).
    packed=True: means2d[nnz,2], radii[nnz], depths[nnz] with n_cameras and camera_ids[nnz] (gaussian_ids is accepted
    and not needed); flatten_ids then index the packed rows.
    periodic=True (not in gsplat; `rasterization` sets it for camera_model="spherical"): the image is periodic in x
    with period tile_width*tile_size, so a footprint that leaves through one side edge is binned into the tile columns
    of the other side (the +-pi seam of an equirectangular panorama).
    conics + opacities (not in gsplat; `rasterization(tile_cull=True)` passes them): EXACT tile culling -- a tile none of
    whose pixel centres can reach alpha = opacity * exp(-sigma) >= 1/255 is left out of the lists.  gsplat files a Gaussian
    under the whole square of half-width ceil(3 sqrt(lambda_max)); the rasteriser then discards such pairs pixel by pixel, so
    the rendered image is bit-identical and only the lists are shorter (about half the entries for small splats)."""
    tile_size = _periodic_tile_size(tile_size, periodic)
    cull = conics is not None and opacities is not None
    if packed:
        return _isect_tiles_packed(means2d, radii, depths, tile_size, tile_width, tile_height, sort, n_cameras,
                                   camera_ids, return_offsets, conics if cull else None, opacities if cull else None)
    C, N = radii.shape
    assert means2d.shape == (C, N, 2), means2d.shape
    assert depths.shape == (C, N), depths.shape
    dev = means2d.device
    means2d, depths = _f32(means2d), _f32(depths)
    radii = radii.to(torch.int32).contiguous()
    M = C * tile_width * tile_height
    tiles_per_gauss = torch.empty(C, N, dtype=torch.int32, device=dev)
    counters = torch.zeros(2 * M + 3, dtype=torch.int32, device=dev)  # counts | cursor (+1) | n | overflow
    tile_counts, cursor = counters[:M], counters[M:2 * M + 1]
    n_isects, overflow = counters[2 * M + 1:2 * M + 2], counters[2 * M + 2:]
    offsets = torch.empty(C, tile_height, tile_width, dtype=torch.int32, device=dev)
    cull_rec = _cull_records(means2d, conics, opacities) if cull else None
    call("so_isect_count", C, N, ptr(means2d), ptr(radii), tile_size, tile_width, tile_height,
         ptr(tiles_per_gauss), ptr(tile_counts), ptr(offsets), ptr(n_isects), ptr(cull_rec), stream())
    total = int(n_isects.item())
    isect_ids = torch.empty(total, dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(total, dtype=torch.int32, device=dev)
    if total > 0:
        if sort:
            check = os.environ.get("SPLAT_ONE_AMD_CHECK_LISTS") == "1"     # debugging aid: did the fill pass write every slot?
            # culled lists: the fill pass repeats the count pass's tile test (bit-identical arithmetic, so_common.hpp); should
            # the two ever disagree, a zero key (Gaussian 0, sorted first) is a harmless slot where garbage would be a fault
            keys = (torch.full((total,), -1, dtype=torch.int64, device=dev) if check
                    else (torch.zeros if cull else torch.empty)(total, dtype=torch.int64, device=dev))
            call("so_isect_fill", C, N, ptr(means2d), ptr(radii), ptr(depths), tile_size, tile_width,
                 tile_height, ptr(offsets), ptr(n_isects), ptr(cursor), total, ptr(keys), ptr(flatten_ids),
                 ptr(isect_ids), ptr(overflow), 0, ptr(cull_rec), stream())
            if check:
                bad = (flatten_ids < 0) | (flatten_ids >= C * N)
                if bool(bad.any()) or int(overflow.item()) != 0:
                    pos = torch.nonzero(bad)[:8, 0].tolist()
                    tiles = [int(torch.searchsorted(offsets.reshape(-1).long(), torch.tensor(p_, device=dev), right=True)) - 1 for p_ in pos]
                    raise RuntimeError(f"isect_tiles: {int(bad.sum())} of {total} list slots were not written by the fill pass "
                                       f"(overflow flag {int(overflow.item())}); first positions {pos} in tiles {tiles}, "
                                       f"ids {flatten_ids[pos].tolist() if pos else []}")
        else:
            assert not cull, "isect_tiles(sort=False) keeps gsplat's lists (no tile culling)"
            cum = torch.cumsum(tiles_per_gauss.reshape(-1).to(torch.int64), 0).contiguous()
            call("so_isect_emit_unsorted", C, N, ptr(means2d), ptr(radii), ptr(depths), ptr(cum), tile_size,
                 tile_width, tile_height, ptr(isect_ids), ptr(flatten_ids), stream())
    if return_offsets:
        return tiles_per_gauss, isect_ids, flatten_ids, offsets
    return tiles_per_gauss, isect_ids, flatten_ids


def _cull_records(means2d: Tensor, conics: Tensor, opacities: Tensor) -> Tensor:
    """64-byte records {x, y, conic, opacity, ...} for the exact tile test of the binning kernels."""
    rows = opacities.numel()
    assert conics.shape == opacities.shape + (3,) and means2d.shape == opacities.shape + (2,), (conics.shape, opacities.shape)
    rec = torch.empty(max(rows, 1), 16, dtype=torch.float32, device=means2d.device)
    call("so_rec_pack", rows, ptr(_f32(means2d)), ptr(_f32(conics.detach())), 0, ptr(_f32(opacities.detach())), ptr(rec), 0,
         stream())
    return rec


def rec_pack_unpack_roundtrip(dev) -> bool:
    """Self-check of `so_rec_pack` / `so_rec_unpack_grads` (the record helpers of `rasterize_to_pixels`): what is packed
    into slots {x, y, conic, opacity, rgb} comes back from the gradient slots {v_x, v_y, v_conic, v_rgb, v_opacity}."""
    n = 1000
    g = torch.Generator(device=dev).manual_seed(3)
    m, cn, col, op = (torch.rand(n, k, generator=g, device=dev) for k in (2, 3, 3, 1))
    rec = torch.empty(n, 16, device=dev)
    call("so_rec_pack", n, ptr(m), ptr(cn), ptr(col), ptr(op), ptr(rec), 0, stream())
    ok = (torch.equal(rec[:, 0:2], m) and torch.equal(rec[:, 2:5], cn) and torch.equal(rec[:, 5:6], op)
          and torch.equal(rec[:, 6:9], col))
    vrec = torch.arange(n * 16, device=dev, dtype=torch.float32).reshape(n, 16).contiguous()
    outs = [torch.empty(n, k, device=dev) for k in (2, 3, 3, 1, 2)]
    call("so_rec_unpack_grads", n, ptr(vrec), *[ptr(o) for o in outs], stream())
    return bool(ok and torch.equal(outs[0], vrec[:, 0:2]) and torch.equal(outs[1], vrec[:, 2:5])
                and torch.equal(outs[2], vrec[:, 5:8]) and torch.equal(outs[3], vrec[:, 8:9]) and torch.equal(outs[4], vrec[:, 9:11]))


def _isect_tiles_packed(means2d, radii, depths, tile_size, tile_width, tile_height, sort, n_cameras, camera_ids,
                        return_offsets, conics=None, opacities=None):
    nnz = radii.shape[0]
    assert means2d.shape == (nnz, 2) and depths.shape == (nnz,), (means2d.shape, depths.shape)
    assert n_cameras is not None and camera_ids is not None and camera_ids.shape == (nnz,), "packed: n_cameras / camera_ids"
    C = int(n_cameras)
    if C == 1:          # one camera: the packed rows ARE a dense [1, nnz] problem
        kw = {"conics": conics[None], "opacities": opacities[None]} if conics is not None else {}
        out = isect_tiles(means2d[None], radii[None], depths[None], tile_size, tile_width, tile_height, sort=sort,
                          return_offsets=return_offsets, **kw)
        return (out[0][0],) + tuple(out[1:])
    # several cameras: the binning kernels take the camera of a row from its position in a [C, n] grid -- give every
    # camera a row of nnz slots and put packed row r into slot (camera_ids[r], r); empty slots have radius 0
    dev = means2d.device
    rows = torch.arange(nnz, device=dev)
    slot = camera_ids.to(torch.int64) * nnz + rows
    g_radii = torch.zeros(C * nnz, dtype=torch.int32, device=dev)
    g_means2d = torch.zeros(C * nnz, 2, dtype=torch.float32, device=dev)
    g_depths = torch.zeros(C * nnz, dtype=torch.float32, device=dev)
    g_radii[slot] = radii.to(torch.int32)
    g_means2d[slot] = means2d.to(torch.float32)
    g_depths[slot] = depths.to(torch.float32)
    cull_kw = {}
    if conics is not None:
        g_conics = torch.zeros(C * nnz, 3, dtype=torch.float32, device=dev)
        g_opac = torch.zeros(C * nnz, dtype=torch.float32, device=dev)
        g_conics[slot] = conics.detach().to(torch.float32)
        g_opac[slot] = opacities.detach().to(torch.float32)
        cull_kw = {"conics": g_conics.view(C, nnz, 3), "opacities": g_opac.view(C, nnz)}
    out = isect_tiles(g_means2d.view(C, nnz, 2), g_radii.view(C, nnz), g_depths.view(C, nnz), tile_size, tile_width,
                      tile_height, sort=sort, return_offsets=return_offsets, **cull_kw)
    tiles_per_gauss = out[0].reshape(-1)[slot]
    flatten_ids = out[2] % max(nnz, 1)                 # slot index -> packed row
    return (tiles_per_gauss, out[1], flatten_ids) + tuple(out[3:])


@torch.no_grad()
def isect_tiles_static(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int,
                       tile_height: int, capacity: int, workspace: Optional[dict] = None,
                       want_isect_ids: bool = False, periodic: bool = False,
                       conics: Optional[Tensor] = None, opacities: Optional[Tensor] = None) -> dict:
    """Sync-free binning into caller-sized buffers (hipGraph-capturable).  Returns a dict with
    tiles_per_gauss, isect_offsets, flatten_ids[capacity], n_isects (device i32[1]), overflow
    (device i32[1]) and optionally isect_ids[capacity].  Nothing is read back to the host.
    periodic, conics + opacities: as in `isect_tiles`."""
    tile_size = _periodic_tile_size(tile_size, periodic)
    cull_rec = _cull_records(means2d, conics, opacities) if (conics is not None and opacities is not None) else None
    C, N = radii.shape
    dev = means2d.device
    M = C * tile_width * tile_height
    ws = workspace if workspace is not None else {}
    def buf(name, shape, dtype):
        t = ws.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
            t = torch.empty(shape, dtype=dtype, device=dev)
            ws[name] = t
        return t
    tiles_per_gauss = buf("tiles_per_gauss", (C, N), torch.int32)
    counters = buf("counters", (2 * M + 3,), torch.int32)
    counters.zero_()
    offsets = buf("isect_offsets", (C, tile_height, tile_width), torch.int32)
    keys = buf("keys", (capacity,), torch.int64)
    flatten_ids = buf("flatten_ids", (capacity,), torch.int32)
    isect_ids = buf("isect_ids", (capacity,), torch.int64) if want_isect_ids else None
    tile_counts, cursor = counters[:M], counters[M:2 * M + 1]
    n_isects, overflow = counters[2 * M + 1:2 * M + 2], counters[2 * M + 2:]
    means2d, depths = _f32(means2d), _f32(depths)
    call("so_isect_count", C, N, ptr(means2d), ptr(radii), tile_size, tile_width, tile_height,
         ptr(tiles_per_gauss), ptr(tile_counts), ptr(offsets), ptr(n_isects), ptr(cull_rec), stream())
    call("so_isect_fill", C, N, ptr(means2d), ptr(radii), ptr(depths), tile_size, tile_width, tile_height,
         ptr(offsets), ptr(n_isects), ptr(cursor), capacity, ptr(keys), ptr(flatten_ids), ptr(isect_ids),
         ptr(overflow), 0, ptr(cull_rec), stream())
    return dict(tiles_per_gauss=tiles_per_gauss, isect_offsets=offsets, flatten_ids=flatten_ids,
                isect_ids=isect_ids, n_isects=n_isects, overflow=overflow)


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """isect_ids[I] (sorted) -> offsets[C,tile_h,tile_w] i32."""
    assert isect_ids.dtype == torch.int64, isect_ids.dtype
    isect_ids = isect_ids.contiguous()
    offsets = torch.empty(n_cameras, tile_height, tile_width, dtype=torch.int32, device=isect_ids.device)
    call("so_isect_offset_encode", isect_ids.numel(), ptr(isect_ids), n_cameras, tile_width, tile_height,
         ptr(offsets), stream())
    return offsets


# ---------------------------------------------------------------------------------------------
# K9/K10 rasterise
# ---------------------------------------------------------------------------------------------
class _RasterizeToPixels(torch.autograd.Function):
    """RGB without tile masks goes through the 64-byte-record kernels (`so_rec_pack` -> `so_rasterize_*_packed` ->
    `so_rec_unpack_grads`): one cache line per list entry for the gathers and ONE atomic record per (quadrant, Gaussian)
    in the backward instead of nine atomics into four arrays (c2: forward 113 -> 70 us, backward 280 -> 150 us).  Other
    channel counts and masked renders use the array kernels.  Same results either way (tests/test_gpu_ops.py)."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size,
                isect_offsets, flatten_ids, n_isects_dev, absgrad):
        C, N = isect_offsets.shape[0], opacities.shape[-1]   # rows are addressed through flatten_ids: [C,N] or packed [nnz]
        D = colors.shape[-1]
        dev = means2d.device
        render_colors = torch.empty(C, height, width, D, dtype=torch.float32, device=dev)
        render_alphas = torch.empty(C, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        n_host = flatten_ids.numel()      # exact count, or (static mode) the capacity that bounds the device count
        rows = opacities.numel()
        rec = None
        if D == 3 and masks is None and rows > 0:
            rec = torch.empty(rows, 16, dtype=torch.float32, device=dev)
            call("so_rec_pack", rows, ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(rec), 0, stream())
            # (C, N) only size the grid and bound the row index: the records are addressed through flatten_ids
            call("so_rasterize_fwd_packed", C, N, width, height, tile_size, ptr(rec), ptr(backgrounds),
                 ptr(isect_offsets), ptr(flatten_ids), ptr(n_isects_dev), n_host, ptr(render_colors),
                 ptr(render_alphas), ptr(last_ids), stream())
            ctx.save_for_backward(rec, backgrounds, isect_offsets, flatten_ids, n_isects_dev, render_alphas, last_ids,
                                  means2d)       # (means2d only carries the `.absgrad` side channel)
            ctx.shapes = (means2d.shape, conics.shape, colors.shape, opacities.shape)
        else:
            call("so_rasterize_fwd", C, N, D, width, height, tile_size, ptr(means2d), ptr(conics), ptr(colors),
                 ptr(opacities), ptr(backgrounds), ptr(masks), ptr(isect_offsets), ptr(flatten_ids),
                 ptr(n_isects_dev), n_host, ptr(render_colors), ptr(render_alphas), ptr(last_ids), stream())
            ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets,
                                  flatten_ids, n_isects_dev, render_alphas, last_ids)
        ctx.records = rec is not None
        ctx.cfg = (width, height, tile_size, absgrad, n_host, C, N, D)
        return render_colors, render_alphas

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        width, height, tile_size, absgrad, n_host, C, N, D = ctx.cfg
        v_render_colors = v_render_colors.contiguous()
        v_render_alphas = v_render_alphas.contiguous()
        if ctx.records:
            (rec, backgrounds, isect_offsets, flatten_ids, n_isects_dev, render_alphas, last_ids,
             means2d) = ctx.saved_tensors
            s_m, s_cn, s_col, s_op = ctx.shapes
            dev = rec.device
            rows = rec.shape[0]
            vrec = torch.zeros(rows, 16, dtype=torch.float32, device=dev)
            call("so_rasterize_bwd_packed", C, N, width, height, tile_size, ptr(rec), ptr(backgrounds),
                 ptr(isect_offsets), ptr(flatten_ids), ptr(n_isects_dev), n_host, ptr(render_alphas), ptr(last_ids),
                 ptr(v_render_colors), ptr(v_render_alphas), ptr(vrec), int(absgrad), stream())
            v_means2d = torch.empty(s_m, dtype=torch.float32, device=dev)
            v_conics = torch.empty(s_cn, dtype=torch.float32, device=dev)
            v_colors = torch.empty(s_col, dtype=torch.float32, device=dev)
            v_opacities = torch.empty(s_op, dtype=torch.float32, device=dev)
            v_abs = torch.empty(s_m, dtype=torch.float32, device=dev) if absgrad else None
            call("so_rec_unpack_grads", rows, ptr(vrec), ptr(v_means2d), ptr(v_conics), ptr(v_colors),
                 ptr(v_opacities), ptr(v_abs), stream())
        else:
            (means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, n_isects_dev,
             render_alphas, last_ids) = ctx.saved_tensors
            v_means2d = torch.zeros_like(means2d)
            v_conics = torch.zeros_like(conics)
            v_colors = torch.zeros_like(colors)
            v_opacities = torch.zeros_like(opacities)
            v_abs = torch.zeros_like(means2d) if absgrad else None
            call("so_rasterize_bwd", C, N, D, width, height, tile_size, ptr(means2d), ptr(conics), ptr(colors),
                 ptr(opacities), ptr(backgrounds), ptr(masks), ptr(isect_offsets), ptr(flatten_ids),
                 ptr(n_isects_dev), n_host, ptr(render_alphas), ptr(last_ids), ptr(v_render_colors),
                 ptr(v_render_alphas), ptr(v_means2d), ptr(v_abs), ptr(v_conics), ptr(v_colors), ptr(v_opacities),
                 stream())
        if absgrad:
            means2d.absgrad = v_abs          # same side channel as gsplat: strategy reads `.absgrad`
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[4]:
            v_bg = (v_render_colors * (1.0 - render_alphas)).sum(dim=(1, 2))
        return v_means2d, v_conics, v_colors, v_opacities, v_bg, None, None, None, None, None, None, None, None


def rasterize_to_pixels(
    means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, image_width: int,
    image_height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
    backgrounds: Optional[Tensor] = None, masks: Optional[Tensor] = None, packed: bool = False,
    absgrad: bool = False, n_isects: Optional[Tensor] = None, periodic: bool = False,
) -> Tuple[Tensor, Tensor]:
    """means2d[C,N,2] conics[C,N,3] colors[C,N,D] opacities[C,N] -> (render_colors[C,H,W,D],
    render_alphas[C,H,W,1]).  `n_isects` (device i32[1]) switches to the static-capacity mode of
    `isect_tiles_static`.  Channel counts outside the compiled set are zero-padded (as gsplat does).
    packed=True: means2d[nnz,2] conics[nnz,3] colors[nnz,D] opacities[nnz], flatten_ids indexing those rows (the
    kernels address Gaussians through flatten_ids only, so both layouts run the same code).
    periodic=True (not in gsplat): the image is periodic in x with period image_width (which tile_size must divide);
    every Gaussian is evaluated at the copy nearest to the tile, matching `isect_tiles(periodic=True)`."""
    C = isect_offsets.shape[0]
    lead = tuple(opacities.shape)
    if packed:
        assert len(lead) == 1, opacities.shape
    else:
        assert len(lead) == 2 and lead[0] == C, opacities.shape
    assert means2d.shape == lead + (2,), means2d.shape
    assert conics.shape == lead + (3,), conics.shape
    assert colors.shape[:-1] == lead, colors.shape
    assert tile_size in (8, 16), f"tile_size {tile_size} not supported (8 or 16)"
    th, tw = isect_offsets.shape[1:]
    assert isect_offsets.shape[0] == C
    assert tw * tile_size >= image_width and th * tile_size >= image_height, "tile grid does not cover the image"
    assert not periodic or image_width % tile_size == 0, "periodic images need image_width % tile_size == 0"
    D = colors.shape[-1]
    if backgrounds is not None:
        assert backgrounds.shape == (C, D), backgrounds.shape
        backgrounds = _f32(backgrounds)
    if masks is not None:
        assert masks.shape == isect_offsets.shape, masks.shape
        masks = masks.to(torch.uint8).contiguous()
    pad = 0
    if D not in SUPPORTED_CHANNELS:
        assert D <= SUPPORTED_CHANNELS[-1], f"too many channels: {D} (chunk them, as `rasterization` does)"
        Dp = min(d for d in SUPPORTED_CHANNELS if d >= D)
        pad = Dp - D
        colors = torch.cat([colors, torch.zeros(lead + (pad,), dtype=colors.dtype, device=colors.device)], -1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, pad, dtype=torch.float32, device=colors.device)], -1)
    rc, ra = _RasterizeToPixels.apply(
        _f32(means2d), _f32(conics), _f32(colors), _f32(opacities), backgrounds, masks, int(image_width),
        int(image_height), _periodic_tile_size(tile_size, periodic), isect_offsets.to(torch.int32).contiguous(),
        flatten_ids.to(torch.int32).contiguous(), n_isects, bool(absgrad))
    if pad:
        rc = rc[..., :D]
    return rc, ra
