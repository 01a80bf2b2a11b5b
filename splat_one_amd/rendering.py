"""`rasterization(...)` -- drop-in for `gsplat.rendering.rasterization` as the reference calls it at
/root/reference/utils/gsplat_utils/gsplat_trainer.py:477-494 (same keyword names, same return triple,
same `meta` keys the densification strategy consumes: SURVEY.md 8b).

Pipeline (all HIP, gfx950): projection (K1) -> SH colour (K4, +0.5, clamp) -> tile binning + per-tile
depth sort (K6-K8) -> tile rasteriser (K9); backward through K10, K5, K2 via torch.autograd.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import raster_op
from .ops import (camera_inverse, sh_view_colors, fully_fused_projection, isect_offset_encode, isect_tiles, isect_tiles_static,
                  rasterize_to_pixels, spherical_harmonics)

RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")
_PENDING = object()
LIST_KEYS = ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets")


class Meta(dict):
    """The `meta` / `info` dict of `rasterization`.  A plain dict for every key the call computes anyway; the per-tile LISTS
    (`tiles_per_gauss`, `isect_ids`, `flatten_ids`, `isect_offsets`) are computed when first asked for -- gsplat's own
    lists, entry for entry (no tile culling, compact layout, global stable order), from the projected centres / radii /
    depths of this call -- because the kernels work on shorter, differently laid-out lists of their own and the
    reference's trainer never reads these keys (gsplat_trainer.py:616-622, 707, 722-724, 750)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._thunks = []

    def set_lazy(self, keys, thunk) -> None:
        """`thunk()` returns the values of `keys` (a tuple), evaluated when one of them is first read"""
        self._thunks.append((tuple(keys), thunk))
        for k in keys:
            dict.__setitem__(self, k, _PENDING)

    def set_lists(self, thunk) -> None:
        self.set_lazy(LIST_KEYS, thunk)

    def _resolve(self, key=None) -> None:
        keep = []
        for keys, thunk in self._thunks:
            if key is not None and key not in keys:
                keep.append((keys, thunk))
                continue
            for k, v in zip(keys, thunk()):
                if dict.__getitem__(self, k) is _PENDING:
                    dict.__setitem__(self, k, v)
        self._thunks = keep

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is _PENDING:
            self._resolve(k)
            v = dict.__getitem__(self, k)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        self._resolve()
        return dict.items(self)

    def values(self):
        self._resolve()
        return dict.values(self)

    def copy(self):
        self._resolve()
        return Meta(dict.copy(self))


def _all_to_all_rows(inp: Tensor, send: list, recv: list) -> Tensor:
    """Variable-split all-to-all along dim 0.  RCCL directly; the gloo backend (tests with several ranks on one GPU)
    stages through host memory."""
    out = torch.empty((sum(recv),) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
    if dist.get_backend() == "nccl":
        dist.all_to_all_single(out, inp.contiguous(), output_split_sizes=recv, input_split_sizes=send)
    else:
        o = torch.empty(out.shape, dtype=inp.dtype)
        dist.all_to_all_single(o, inp.detach().cpu().contiguous(), output_split_sizes=recv, input_split_sizes=send)
        out.copy_(o)
    return out


class _AllToAllRows(torch.autograd.Function):
    """Differentiable all-to-all: the gradient of what a rank received travels back to the rank that sent it."""

    @staticmethod
    def forward(ctx, inp, send, recv):
        ctx.splits = (list(send), list(recv))
        return _all_to_all_rows(inp, list(send), list(recv))

    @staticmethod
    def backward(ctx, v_out):
        send, recv = ctx.splits
        return _all_to_all_rows(v_out.contiguous(), recv, send), None, None


@torch.no_grad()
def _gather_cameras(N: int, viewmats: Tensor, Ks: Tensor):
    """(Gaussians per rank, viewmats of all ranks [C_world,4,4], Ks of all ranks) -- every rank brings the same number
    of cameras (as gsplat requires)."""
    assert dist.is_available() and dist.is_initialized(), "distributed=True needs an initialised process group"
    world = dist.get_world_size()
    dev = viewmats.device
    gloo = dist.get_backend() != "nccl"
    mine = torch.cat([torch.tensor([float(N)], dtype=torch.float64), viewmats.detach().double().reshape(-1).cpu(),
                      Ks.detach().double().reshape(-1).cpu()])
    mine = mine if gloo else mine.to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    C = viewmats.shape[0]
    parts = [p.cpu() for p in parts]
    assert all(p.numel() == mine.numel() for p in parts), "every rank must bring the same number of cameras"
    N_world = [int(p[0].item()) for p in parts]
    vm = torch.cat([p[1:1 + 16 * C].reshape(C, 4, 4) for p in parts]).to(device=dev, dtype=torch.float32)
    ks = torch.cat([p[1 + 16 * C:].reshape(C, 3, 3) for p in parts]).to(device=dev, dtype=torch.float32)
    return N_world, vm, ks


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]
    scales: Tensor,  # [N, 3]
    opacities: Tensor,  # [N]
    colors: Tensor,  # [(C,) N, D] or [(C,) N, K, 3]
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: str = "pinhole",
    covars: Optional[Tensor] = None,
    isect_capacity: Optional[int] = None,
    workspace: Optional[dict] = None,
    tile_cull: bool = True,
    fused: Optional[bool] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterise N Gaussians to C cameras.  Returns (render_colors[C,H,W,X], render_alphas[C,H,W,1], meta).

    Differences from the gsplat call, all explicit:
      * `fused` (None = whenever possible): the COMMON SHAPE of the call -- `packed=False`, SH coefficients [N,K,3] shared by
        the cameras, `render_mode="RGB"`, poses without gradient, no `covars` / `distributed` / `sparse_grad`: exactly what
        the reference passes at gsplat_trainer.py:478-493 -- runs as ONE library call each way (`splat_one_amd.raster_op`:
        so_rasterization_fwd / _bwd), with nothing read back to the host.  Any other shape, or `fused=False`, composes the
        operators of `splat_one_amd.ops` as gsplat composes its own.  Same images, alphas, `meta` and gradients either way.
      * `tile_cull=True` (default): the kernels' INTERNAL per-tile lists leave out (Gaussian, tile) pairs that provably reach
        no pixel with alpha >= 1/255.  Images are bit-identical to the un-culled call and gradients equal up to the order of
        the float atomics.  `meta["tiles_per_gauss"]`, `meta["isect_ids"]`, `meta["flatten_ids"]`, `meta["isect_offsets"]`
        are gsplat's lists regardless (computed when first read, class `Meta`).
      * `packed=True` (gsplat's default; the reference passes `Config.packed` = False, gsplat_trainer.py:133, 487):
        every per-Gaussian intermediate and `meta` entry has one row per (camera, Gaussian) pair with a positive
        radius, camera-major, with `meta["camera_ids"]` / `meta["gaussian_ids"]` naming the pair; the rasteriser
        kernels address rows through `flatten_ids`, so both layouts run the same kernels and give the same images.
        `sparse_grad=True` (packed only) returns row-sparse COO gradients for means / quats / scales.
        `isect_capacity` (the sync-free binning) does not apply to packed calls: nnz is read on the host anyway.
      * `distributed=True` (the reference passes `distributed=self.world_size > 1`, gsplat_trainer.py:490): every rank
        holds a shard of the Gaussians and the same number of cameras; the cameras are all-gathered, the shard is
        projected into ALL of them, and ONE differentiable all-to-all of the projected rows {radius, centre, depth, conic,
        opacity, colour} (RCCL; its backward is the reverse all-to-all) gives every rank all Gaussians for its own
        cameras.  `meta` keeps the shard-local pre-exchange tensors ([C_world, N_local], what a per-rank strategy needs),
        `meta["n_cameras"]` the local count.  With `packed=True` only the visible (camera, Gaussian) rows travel (their
        per-destination counts are exchanged first) and the rank rasterises the received packed rows; viewmats / Ks are
        gathered without a gradient path.  (`splat_one_amd.sharded.ShardedEngine` is the fused form of the same scheme.)
      * `isect_capacity` / `workspace` (extensions): preallocated intersection buffers make the call
        free of host synchronisation (hipGraph-capturable); `meta["n_isects"]` then lives on the device.
    """
    meta: Dict = Meta()
    N = means.shape[0]
    C = viewmats.shape[0]
    device = means.device
    assert means.shape == (N, 3), means.shape
    if covars is None:
        assert quats.shape == (N, 4), quats.shape
        assert scales.shape == (N, 3), scales.shape
    else:
        assert covars.shape == (N, 3, 3), covars.shape
        quats, scales = None, None
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert render_mode in RENDER_MODES, render_mode
    assert rasterize_mode in ("classic", "antialiased"), rasterize_mode
    if distributed:
        assert isect_capacity is None, "distributed=True: the exchange sizes are read on the host anyway"
        N_world, viewmats, Ks = _gather_cameras(N, viewmats, Ks)
        C_local, C = C, viewmats.shape[0]
    assert packed or not sparse_grad, "sparse_grad requires packed=True"

    if sh_degree is None:
        # treat colors as post-activation values, [N, D] or [C, N, D]
        assert (colors.dim() == 2 and colors.shape[0] == N) or (colors.dim() == 3 and colors.shape[:2] == (C, N)), colors.shape
    else:
        # treat colors as SH coefficients, [N, K, 3] or [C, N, K, 3]
        assert (colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3) or (
            colors.dim() == 4 and colors.shape[:2] == (C, N) and colors.shape[3] == 3), colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[-2], colors.shape

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    periodic = camera_model == "spherical" and width % tile_size == 0

    def gsplat_lists(means2d_, radii_, depths_, camera_ids_=None, gaussian_ids_=None, n_cam=C):
        """thunk for Meta: gsplat's un-culled compact lists of this call"""
        def thunk():
            with torch.no_grad():
                return isect_tiles(means2d_.detach().contiguous(), radii_.contiguous(), depths_.detach().contiguous(), tile_size,
                                   tile_width, tile_height, packed=packed, n_cameras=n_cam, camera_ids=camera_ids_,
                                   gaussian_ids=gaussian_ids_, return_offsets=True, periodic=periodic)
        return thunk

    if fused is None or fused:
        ok = raster_op.usable(means, quats, scales, opacities, colors, viewmats, Ks, sh_degree=sh_degree, packed=packed,
                              tile_size=tile_size, render_mode=render_mode, sparse_grad=sparse_grad, distributed=distributed,
                              covars=covars, isect_capacity=isect_capacity, backgrounds=backgrounds, camera_model=camera_model)
        assert ok or not fused, "rasterization(fused=True): this call does not have the common shape (see the docstring)"
        if ok:
            return _one_call(means, quats, scales, opacities, colors, None, viewmats, Ks, width, height, sh_degree=sh_degree,
                             near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, tile_size=tile_size,
                             backgrounds=backgrounds, absgrad=absgrad, rasterize_mode=rasterize_mode, camera_model=camera_model,
                             tile_cull=tile_cull)

    # K1 projection
    proj = fully_fused_projection(
        means, covars, quats, scales, viewmats, Ks, width, height, eps2d=eps2d, packed=packed,
        near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, sparse_grad=sparse_grad,
        calc_compensations=(rasterize_mode == "antialiased"), camera_model=camera_model)
    if packed:
        camera_ids, gaussian_ids, radii, means2d, depths, conics, compensations = proj
        opacities = opacities[gaussian_ids]                      # [nnz]
        isect_capacity = None
    else:
        camera_ids = gaussian_ids = None
        radii, means2d, depths, conics, compensations = proj
        opacities = opacities[None, :].expand(C, N)
    if compensations is not None:
        opacities = opacities * compensations
    opacities = opacities.contiguous()

    meta.update({"camera_ids": camera_ids, "gaussian_ids": gaussian_ids, "radii": radii, "means2d": means2d,
                 "depths": depths, "conics": conics, "opacities": opacities})

    # K4 colours
    if sh_degree is None:
        if packed:
            colors = colors[gaussian_ids] if colors.dim() == 2 else colors[camera_ids, gaussian_ids]    # [nnz, D]
        elif colors.dim() == 2:
            colors = colors[None].expand(C, -1, -1)
    elif (not packed and not distributed and colors.dim() == 3 and not viewmats.requires_grad
          and radii.dtype == torch.int32 and tuple(radii.shape) == (C, N)):
        # the common call (dense layout, coefficients shared by the cameras, fixed poses): view directions, SH evaluation,
        # "+ 0.5" and the clamp in one launch each way
        campos = camera_inverse(viewmats)[:, :3, 3].contiguous()
        colors = sh_view_colors(sh_degree, means, campos, colors, radii)
    else:
        camtoworlds = camera_inverse(viewmats)      # (no device synchronisation, unlike torch.inverse)
        if packed:
            dirs = means[gaussian_ids] - camtoworlds[camera_ids, :3, 3]  # [nnz, 3]
            shs = colors[gaussian_ids] if colors.dim() == 3 else colors[camera_ids, gaussian_ids]     # [nnz, K, 3]
        else:
            dirs = means[None, :, :] - camtoworlds[:, None, :3, 3]  # [C, N, 3]
            shs = colors[None].expand(C, -1, -1, -1) if colors.dim() == 3 else colors
        masks = radii > 0
        colors = spherical_harmonics(sh_degree, dirs, shs, masks=masks)  # [C, N, 3] | [nnz, 3]
        colors = torch.clamp_min(colors + 0.5, 0.0)

    if render_mode in ("RGB+D", "RGB+ED"):
        colors = torch.cat((colors, depths[..., None]), dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(backgrounds.shape[0], 1, device=device)], dim=-1)
    elif render_mode in ("D", "ED"):
        colors = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(backgrounds.shape[0], 1, device=device)

    if distributed and packed:
        # packed rows are camera-major over the cameras of ALL ranks: the rows of the cameras rank j owns go to rank j
        world, rank = len(N_world), dist.get_rank()
        owner = torch.div(camera_ids, C_local, rounding_mode="floor")
        send = torch.bincount(owner, minlength=world)
        every = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(every, send)
        send_counts = [int(v) for v in send.tolist()]
        recv_counts = [int(e[rank]) for e in every]
        as_f = lambda ids: ids.to(torch.int32).view(torch.float32)[:, None]         # bit-preserving through the exchange
        rec = torch.cat([as_f(radii), as_f(camera_ids - owner * C_local), as_f(gaussian_ids + sum(N_world[:rank])),
                         means2d, depths[:, None], conics, opacities[:, None], colors], dim=-1)
        rows = _AllToAllRows.apply(rec, send_counts, recv_counts)
        C, N = C_local, sum(N_world)
        radii = rows[:, 0].contiguous().view(torch.int32)
        camera_ids = rows[:, 1].contiguous().view(torch.int32).long()
        gaussian_ids = rows[:, 2].contiguous().view(torch.int32).long()
        means2d, depths, conics = rows[:, 3:5].contiguous(), rows[:, 5].contiguous(), rows[:, 6:9].contiguous()
        opacities, colors = rows[:, 9].contiguous(), rows[:, 10:]
    elif distributed:
        # one all-to-all of the projected rows: block c of the shard's [C_world, N_local] grid goes to the rank owning camera c
        F = 8 + colors.shape[-1]
        rec = torch.cat([radii.view(torch.float32)[..., None], means2d, depths[..., None], conics, opacities[..., None],
                         colors], dim=-1).reshape(C * N, F)
        world = len(N_world)
        got = _AllToAllRows.apply(rec, [C_local * N] * world, [C_local * n for n in N_world])
        rows = torch.cat([b.view(C_local, n, F) for b, n in zip(got.split([C_local * n for n in N_world]), N_world)], dim=1)
        C, N = C_local, sum(N_world)
        radii = rows[..., 0].contiguous().view(torch.int32)
        means2d, depths, conics = rows[..., 1:3].contiguous(), rows[..., 3].contiguous(), rows[..., 4:7].contiguous()
        opacities, colors = rows[..., 7].contiguous(), rows[..., 8:]

    # K6-K8 binning + sort + offsets
    n_isects_dev = None
    # (periodic: an equirectangular panorama is periodic in x -- footprints continue across the +-pi seam when the tile grid lines up)
    cull_kw = {"conics": conics, "opacities": opacities} if tile_cull else {}
    if isect_capacity is None:
        tiles_per_gauss, isect_ids, flatten_ids, isect_offsets = isect_tiles(
            means2d, radii, depths, tile_size, tile_width, tile_height, packed=packed, n_cameras=C,
            camera_ids=camera_ids, gaussian_ids=gaussian_ids, return_offsets=True, periodic=periodic, **cull_kw)
    else:
        st = isect_tiles_static(means2d, radii, depths, tile_size, tile_width, tile_height, int(isect_capacity),
                                workspace=workspace, want_isect_ids=False, periodic=periodic, **cull_kw)
        tiles_per_gauss, isect_ids, flatten_ids, isect_offsets = (
            st["tiles_per_gauss"], st["isect_ids"], st["flatten_ids"], st["isect_offsets"])
        n_isects_dev = st["n_isects"]
        meta.update({"n_isects": st["n_isects"], "isect_overflow": st["overflow"]})

    meta.update({"tile_width": tile_width, "tile_height": tile_height, "tiles_per_gauss": tiles_per_gauss,
                 "isect_ids": isect_ids, "flatten_ids": flatten_ids, "isect_offsets": isect_offsets,
                 "width": width, "height": height, "tile_size": tile_size, "n_cameras": C})
    meta["n_isects_kernel"] = n_isects_dev if n_isects_dev is not None else torch.tensor(flatten_ids.numel(), device=device)
    if tile_cull and isect_capacity is None:
        # the rasteriser below walks the culled lists; what `meta` shows are gsplat's (computed if somebody reads them)
        meta.set_lists(gsplat_lists(means2d, radii, depths, camera_ids, gaussian_ids, C))

    # K9 rasterise (channel-chunked like gsplat when D > channel_chunk)
    colors = colors.contiguous()
    if colors.shape[-1] > channel_chunk:
        n_chunks = (colors.shape[-1] + channel_chunk - 1) // channel_chunk
        render_colors, render_alphas = [], []
        for i in range(n_chunks):
            colors_chunk = colors[..., i * channel_chunk:(i + 1) * channel_chunk].contiguous()
            bg_chunk = backgrounds[..., i * channel_chunk:(i + 1) * channel_chunk] if backgrounds is not None else None
            rc, ra = rasterize_to_pixels(means2d, conics, colors_chunk, opacities, width, height, tile_size,
                                         isect_offsets, flatten_ids, backgrounds=bg_chunk, packed=packed,
                                         absgrad=absgrad, n_isects=n_isects_dev, periodic=periodic)
            render_colors.append(rc)
            render_alphas.append(ra)
        render_colors = torch.cat(render_colors, dim=-1)
        render_alphas = render_alphas[0]
    else:
        render_colors, render_alphas = rasterize_to_pixels(
            means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids,
            backgrounds=backgrounds, packed=packed, absgrad=absgrad, n_isects=n_isects_dev, periodic=periodic)
    if render_mode in ("ED", "RGB+ED"):
        # normalise the accumulated depth to get the expected depth
        render_colors = torch.cat(
            [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
    return render_colors, render_alphas, meta


def _one_call(means, quats, scales, opacities, colors, shN, viewmats, Ks, width, height, *, sh_degree, near_plane, far_plane,
              radius_clip, eps2d, tile_size, backgrounds, absgrad, rasterize_mode, camera_model, tile_cull):
    """The common shape through `raster_op` (one library call each way) + the `meta` dict around its outputs.  shN given:
    scales / opacities / colors are the raw parameters (log-scales, logits, sh0)."""
    C = viewmats.shape[0]
    tile_width, tile_height = math.ceil(width / float(tile_size)), math.ceil(height / float(tile_size))
    periodic = camera_model == "spherical" and width % tile_size == 0
    (render_colors, render_alphas, means2d, radii, depths, conics, opac_v, _rgb, kernel_isects) = raster_op.fused_rasterization(
        means, quats, scales, opacities, colors, viewmats, Ks, width, height, sh_degree=sh_degree, near_plane=near_plane,
        far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, tile_size=tile_size, backgrounds=backgrounds, absgrad=absgrad,
        antialiased=(rasterize_mode == "antialiased"), camera_model=camera_model, tile_cull=tile_cull, shN=shN)
    meta = Meta({"camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths,
                 "conics": conics, "opacities": opac_v, "tile_width": tile_width, "tile_height": tile_height,
                 "width": width, "height": height, "tile_size": tile_size, "n_cameras": C})

    def gsplat_lists():
        with torch.no_grad():
            return isect_tiles(means2d.detach().contiguous(), radii.contiguous(), depths.detach().contiguous(), tile_size, tile_width,
                               tile_height, packed=False, n_cameras=C, return_offsets=True, periodic=periodic)
    meta.set_lists(gsplat_lists)
    # (extension) how many (Gaussian, tile) pairs the kernels' own -- culled, binned -- lists held: a device scalar
    meta.set_lazy(("n_isects_kernel",), lambda: (kernel_isects(),))
    return render_colors, render_alphas, meta


def rasterization_from_parameters(means: Tensor, quats: Tensor, log_scales: Tensor, logit_opacities: Tensor, sh0: Tensor,
                                  shN: Tensor, viewmats: Tensor, Ks: Tensor, width: int, height: int, **kw):
    """What `Runner.rasterize_splats` computes at /root/reference/utils/gsplat_utils/gsplat_trainer.py:456-494,

        rasterization(means, quats, exp(log_scales), sigmoid(logit_opacities), cat([sh0, shN], 1), viewmats, Ks, ...),

    for a caller that HOLDS the raw parameters: when the call has the common shape (see `rasterization`) the activations
    and the concatenation run inside the kernels -- no exp / sigmoid / cat launches, no autograd nodes for them, the
    coefficient gradient lands in `sh0.grad` / `shN.grad` without being assembled and split again.  Any other shape (or
    fused=False) evaluates exactly the expression above.  Same return triple."""
    fused = kw.pop("fused", None)
    d = dict(near_plane=0.01, far_plane=1e10, radius_clip=0.0, eps2d=0.3, sh_degree=None, packed=True, tile_size=16, backgrounds=None,
             render_mode="RGB", sparse_grad=False, absgrad=False, rasterize_mode="classic", distributed=False,
             camera_model="pinhole", covars=None, isect_capacity=None, tile_cull=True)
    d.update({k: v for k, v in kw.items() if k in d})
    ok = (fused is None or fused) and shN.dim() == 3 and sh0.dim() == 3 and sh0.shape[1] == 1 and shN.shape[0] == sh0.shape[0] \
        and raster_op.usable(means, quats, log_scales, logit_opacities, sh0, viewmats, Ks, sh_degree=d["sh_degree"], packed=d["packed"],
                             tile_size=d["tile_size"], render_mode=d["render_mode"], sparse_grad=d["sparse_grad"],
                             distributed=d["distributed"], covars=d["covars"], isect_capacity=d["isect_capacity"],
                             backgrounds=d["backgrounds"], camera_model=d["camera_model"]) \
        and shN.dtype == torch.float32 and (d["sh_degree"] + 1) ** 2 <= 1 + shN.shape[1]
    if not ok:
        return rasterization(means, quats, torch.exp(log_scales), torch.sigmoid(logit_opacities), torch.cat([sh0, shN], 1),
                             viewmats, Ks, width, height, fused=fused, **kw)
    N, C = means.shape[0], viewmats.shape[0]
    assert means.shape == (N, 3) and quats.shape == (N, 4) and log_scales.shape == (N, 3) and logit_opacities.shape == (N,), \
        (means.shape, quats.shape, log_scales.shape, logit_opacities.shape)
    assert viewmats.shape == (C, 4, 4) and Ks.shape == (C, 3, 3), (viewmats.shape, Ks.shape)
    assert d["rasterize_mode"] in ("classic", "antialiased"), d["rasterize_mode"]
    return _one_call(means, quats, log_scales, logit_opacities, sh0, shN, viewmats, Ks, width, height, sh_degree=d["sh_degree"],
                     near_plane=d["near_plane"], far_plane=d["far_plane"], radius_clip=d["radius_clip"], eps2d=d["eps2d"],
                     tile_size=d["tile_size"], backgrounds=d["backgrounds"], absgrad=d["absgrad"],
                     rasterize_mode=d["rasterize_mode"], camera_model=d["camera_model"], tile_cull=d["tile_cull"])
