"""`gsplat.rendering.rasterization` WHOLE as one autograd.Function over two C-ABI calls.

The reference renders through one call (/root/reference/utils/gsplat_utils/gsplat_trainer.py:477-494) and
differentiates it with `loss.backward()` (:655).  For the common shape of that call -- dense layout, SH coefficients
[N,K,3] shared by the cameras, fixed poses, "RGB" -- `fused_rasterization` runs `so_rasterization_fwd` /
`so_rasterization_bwd` (include/splat_one_amd.h, csrc/raster_op.hip): projection + SH colour + binned tile lists +
per-tile sort + rasteriser in ONE library call, their backward in one more, nothing read back to the host (the
operator-by-operator path reads the intersection count once per call to size its lists, which keeps the host from
running ahead of the GPU).  `rendering.rasterization` dispatches here; every other shape takes the operators of `ops.py`.

What the call returns is what gsplat returns: images, alphas and `info` with `means2d` a graph tensor whose `.grad` /
`.absgrad` the densification strategy reads after the backward (gsplat_trainer.py:616-622, 744-752).

Per-tile lists are BINNED (fixed slots per tile, include/splat_one_amd.h so_isect_sort_bins).  The slot count is measured
-- one synchronisation -- as 8x the fullest tile on the first call of a (device, stream, tile grid) and again whenever the
call's Gaussian count has grown 1.5x past the last measurement (a denser checkpoint at the same resolution), and followed
from then on through a host-mapped status word each forward publishes {fullest tile, overflow}: the next call reads it
without synchronising and enlarges the bins while they are still half empty.  What an overflow can do:
  * a call that needs NO gradient (eval, viewer frames, `torch.no_grad()`) waits for its own status word and runs again on
    larger bins: it never returns an image rendered from cut lists;
  * a call that will be differentiated does not wait (the host keeps running ahead of the GPU).  If the fullest tile more
    than doubled since the previous call, that tile is rendered from the `slots` entries that arrived first (arrival order:
    NOT the front-most ones), and the backward of that call returns EXACTLY ZERO gradients -- the rasteriser's backward
    kernel reads the overflow flag on the device (so_raster_desc, LossFinal.skip) -- so nothing is learnt from the cut
    image; the following call reports it with a RuntimeWarning and enlarges the bins, `pending_overflow()` lets a training
    loop ask after its last call;
  * `SPLAT_ONE_AMD_EXACT_LISTS=1` (or `rasterization(..., fused=False)`) selects the exact-size operator path (gsplat's own
    shape: one host read per call).
"""
from __future__ import annotations

import ctypes
import math
import os
import threading
import warnings
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .list_policy import pick_raster_impl, pick_tile_order
from .ops import camera_model_code

_MIN_SLOTS = 1024


class _Bins:
    """Bin sizing state + the key scratch of one (device, stream, tile grid)."""

    def __init__(self, device: torch.device, M: int):
        self.device, self.M = device, M
        self.lock = threading.Lock()
        # replicated bin counters on images of few tiles (so_raster_desc.bin_replicas; the rule of FusedEngine._pick_bin_replicas):
        # the returning atomics of ONE counter serialise, eight copies of the counters run the binning pass ~4x faster there
        import os
        env = os.environ.get("SPLAT_ONE_AMD_BIN_REPLICAS")
        from .list_policy import pick_bin_replicas
        self.replicas = max(1, min(64, int(env))) if env else pick_bin_replicas(M)
        self.sub_counts = torch.zeros(self.replicas * M + 1, dtype=torch.int32, device=device) if self.replicas > 1 else None
        R = self.replicas
        self.limit = int(max(16, min((2 ** 31 - 1) // M, int(32e9) // (12 * M)))) // R * R
        self.slots = int(min(_MIN_SLOTS, self.limit)) // R * R
        self.key_buf: Optional[Tensor] = None
        self.status = torch.zeros(4, dtype=torch.int32).pin_memory()
        self.status_np = self.status.numpy()         # (the same host-mapped words, read without a torch dispatch)
        self.status_ptr = self.status.data_ptr()
        self.seq = 0
        self.probed = False
        self.n_probe = 0                             # Gaussians per view of the call the bins were last measured on
        self.zero_alphas: Dict[Tuple[int, int, int], Tensor] = {}
        self.overflows = 0
        self.mean_list = 0.0                         # list entries per tile of the last forward whose status has been looked at
        self.fullest = 0                             # ... and its fullest tile
        self.last_order: Optional[Tensor] = None     # the workgroup -> tile table of the last forward, if it built one
        self.last_used = 0                           # _bins_for's clock at the last call on this grid (eviction order)

    def keys(self) -> Tensor:
        n = self.M * self.slots
        if self.key_buf is None or self.key_buf.numel() < n:
            self.key_buf = torch.empty(n, dtype=torch.int64, device=self.device)
        return self.key_buf

    def grow_to(self, fullest: int, factor: int) -> None:
        want = -(-int(factor * fullest + 16) // 256) * 256
        self.slots = int(min(max(self.slots, want), self.limit)) // self.replicas * self.replicas

    def look_at_previous(self) -> None:
        """The status word of the last forward issued on these bins, if that forward has finished (no synchronisation: a
        forward still in flight is simply looked at by a later call)."""
        st = self.status_np
        if self.seq == 0 or int(st[2]) != self.seq:
            return
        fullest, overflow = int(st[0]), int(st[1])
        self.mean_list, self.fullest = float(int(st[3])) / self.M, fullest
        if overflow:
            self.overflows += 1
            warnings.warn(f"splat_one_amd.rasterization: the previous call put {fullest} Gaussians over one tile, more than its "
                          f"{self.slots} list slots -- that tile was rendered from the {self.slots} entries that arrived first "
                          "and the call's backward returned zero gradients (nothing was learnt from the cut image); the bins "
                          "are enlarged now (SPLAT_ONE_AMD_EXACT_LISTS=1 selects exact-size lists at the price of one host "
                          "synchronisation per call)", RuntimeWarning)
            if self.slots >= self.limit:
                raise RuntimeError(f"{fullest} Gaussians over one tile exceed the largest bin this image size allows "
                                   f"({self.limit} slots): call rasterization(..., fused=False)")
            self.grow_to(fullest, 4)
        elif 2 * fullest > self.slots:
            self.grow_to(fullest, 4)


_BINS: Dict[tuple, _Bins] = {}
_BINS_LOCK = threading.Lock()
_MAX_BINS = 64
_BINS_CLOCK = 0


def pending_overflow(synchronize: bool = True) -> int:
    """How many of the LAST calls on any tile grid cut a list (0 or more)?  A differentiated call reports its overflow
    through the call that follows; a training loop asks here after its last iteration.  Synchronises the device by
    default (the status words are written by the forward's last kernel)."""
    if synchronize and torch.cuda.is_available():
        torch.cuda.synchronize()
    n = 0
    for b in list(_BINS.values()):
        st = b.status_np
        if b.seq and int(st[2]) == b.seq and int(st[1]):
            n += 1
    return n


def _bins_for(device: torch.device, M: int, grid: Tuple[int, int, int]) -> _Bins:
    """The bins of one (device, stream, C x tile_w x tile_h) -- the GRID and not only its tile count M: two image shapes of equal
    M (one 145 x 33 view and two 71 x 41 views are 30 tiles each) must not inherit each other's measured list lengths (the second
    would skip its measuring call and cut a 1400-entry list at the first's 1024 slots: fuzz seeds 2222 -> 2344, round 5)."""
    global _BINS_CLOCK
    key = (device.index if device.index is not None else torch.cuda.current_device(), _lib.stream(), M, tuple(grid))
    b = _BINS.get(key)
    if b is None:
        with _BINS_LOCK:
            b = _BINS.get(key)
            if b is None:
                # a viewer that follows its window renders at many sizes, and a grid's key scratch is M x slots x 8 bytes (67 MB at
                # 1080p): keep the _MAX_BINS grids used last (a call in flight holds its own reference; an evicted grid is simply
                # measured again on its next call)
                while len(_BINS) >= _MAX_BINS:
                    del _BINS[min(_BINS, key=lambda k: _BINS[k].last_used)]
                b = _BINS[key] = _Bins(device, M)
    _BINS_CLOCK += 1
    b.last_used = _BINS_CLOCK
    return b


def usable(means: Tensor, quats, scales, opacities: Tensor, colors: Tensor, viewmats: Tensor, Ks: Tensor, *, sh_degree, packed,
           tile_size, render_mode, sparse_grad, distributed, covars, isect_capacity, backgrounds, camera_model) -> bool:
    """Is this the common shape of the call (module docstring)?"""
    if os.environ.get("SPLAT_ONE_AMD_EXACT_LISTS") == "1":
        return False
    if packed or distributed or sparse_grad or covars is not None or sh_degree is None or isect_capacity is not None:
        return False
    if render_mode != "RGB" or tile_size not in (8, 16) or not isinstance(camera_model, str):
        return False
    if colors.dim() != 3 or colors.shape[-1] != 3 or means.shape[0] == 0 or not means.is_cuda:
        return False
    if viewmats.requires_grad or Ks.requires_grad or (backgrounds is not None and backgrounds.shape[-1] != 3):
        return False
    C, N = viewmats.shape[0], means.shape[0]
    if C * N >= 2 ** 31 or any(t.dtype != torch.float32 for t in (means, quats, scales, opacities, colors, viewmats, Ks)):
        return False
    return True


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous()


def _p(t: Optional[Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


_TEMPLATES: Dict[tuple, "_lib.RasterDesc"] = {}


def _desc(cfg: dict, C: int, N: int, K: int) -> "_lib.RasterDesc":
    """A descriptor with every non-pointer field of this call shape filled in (a copy of a cached template: the constant
    half of the 46 fields costs one memcpy instead of 18 attribute stores per call)."""
    key = (cfg["key"], C, N, K)
    t = _TEMPLATES.get(key)
    if t is None:
        if len(_TEMPLATES) > 256:
            _TEMPLATES.clear()
        t = _lib.RasterDesc()
        t.abi_size = ctypes.sizeof(_lib.RasterDesc)
        t.C, t.N, t.K, t.width, t.height, t.tile_size = C, N, K, cfg["width"], cfg["height"], cfg["tile_size"]
        t.sh_degree, t.camera_model = cfg["sh_degree"], camera_model_code(cfg["camera_model"], C)
        t.antialiased, t.absgrad, t.tile_cull = int(cfg["antialiased"]), int(cfg["absgrad"]), int(cfg["tile_cull"])
        t.activated = int(cfg["activated"])
        t.eps2d, t.near_plane, t.far_plane, t.radius_clip = cfg["eps2d"], cfg["near_plane"], cfg["far_plane"], cfg["radius_clip"]
        _TEMPLATES[key] = t
    return _lib.RasterDesc.from_buffer_copy(t)


class _Rasterization(torch.autograd.Function):
    """inputs: means, quats, scales, opacities, sh0, shN -- cfg["activated"]: post-activation scales / opacities and ONE
    coefficient tensor [N,K,3] in `sh0` (shN None: the gsplat call); else the raw parameters (log-scales, opacity logits,
    sh0 [N,1,3], shN [N,K-1,3]: `Runner.rasterize_splats`, which owns the parameters and skips the exp / sigmoid / cat
    round trip through autograd)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, sh0, shN, viewmats, Ks, backgrounds, cfg, holder):
        W, H, ts = cfg["width"], cfg["height"], cfg["tile_size"]
        C, N = viewmats.shape[0], means.shape[0]
        K = sh0.shape[1] if shN is None else 1 + shN.shape[1]
        dev = means.device
        M = C * (-(-W // ts)) * (-(-H // ts))
        need_bwd = any(ctx.needs_input_grad[:6]) or ctx.needs_input_grad[8]
        ctx.set_materialize_grads(False)
        bins = _bins_for(dev, M, (C, -(-W // ts), -(-H // ts)))
        f32, i32 = torch.float32, torch.int32
        d = _desc(cfg, C, N, K)
        d.means, d.quats, d.scales, d.opacities, d.sh0, d.shN = (means.data_ptr(), quats.data_ptr(), scales.data_ptr(),
                                                                   opacities.data_ptr(), sh0.data_ptr(), _p(shN))
        d.viewmats, d.Ks, d.backgrounds = viewmats.data_ptr(), Ks.data_ptr(), _p(backgrounds)
        rec = torch.empty(C * N, 16, dtype=f32, device=dev)
        vrec = torch.empty(C * N, 16, dtype=f32, device=dev) if need_bwd else None
        render_colors = torch.empty(C, H, W, 3, dtype=f32, device=dev)
        render_alphas = torch.empty(C, H, W, 1, dtype=f32, device=dev)
        last_ids = torch.empty(C, H, W, dtype=i32, device=dev)
        counters = torch.empty(2 * M + 3, dtype=i32, device=dev)          # zeroed by the call
        d.rec, d.vrec, d.counters = rec.data_ptr(), _p(vrec), counters.data_ptr()
        d.render_colors, d.render_alphas, d.last_ids = render_colors.data_ptr(), render_alphas.data_ptr(), last_ids.data_ptr()
        d.status_out = bins.status_ptr
        with bins.lock:                      # (the key scratch and the status word are shared by the calls on this stream)
            bins.look_at_previous()
            while True:
                if bins.replicas > 1:        # (every slice of a bin the same size)
                    bins.slots = max(bins.replicas, bins.slots // bins.replicas * bins.replicas)
                flatten_ids = torch.empty(M * bins.slots, dtype=i32, device=dev)
                bins.seq = (bins.seq + 1) & 0x3FFFFFFF
                d.bin_capacity, d.seq = bins.slots, bins.seq
                d.key_buf, d.flatten_ids = bins.keys().data_ptr(), flatten_ids.data_ptr()
                if bins.replicas > 1:
                    d.bin_replicas, d.bin_sub_counts = bins.replicas, bins.sub_counts.data_ptr()
                # longest list first for both rasterisers where the lists are long or uneven (the fused engine's rule, list_policy.py;
                # decided from the lists of the last call whose status word has been looked at -- no read here)
                tile_order = None
                if cfg["tile_size"] == 16 and pick_tile_order(False, 0, bins.mean_list, bins.fullest):
                    tile_order = torch.empty(M, dtype=i32, device=dev)
                    d.tile_order = tile_order.data_ptr()
                bins.last_order = tile_order          # (kept for tools/dbg_order_small.py; replaced by the next call)
                _lib.call("so_rasterization_fwd", ctypes.byref(d), _lib.stream())
                # measure (ONE synchronisation): the first call on this tile grid, a model 1.5x denser than the one the bins
                # were sized on, and every call that needs no gradient (eval / viewer: exact, see the module docstring)
                measure = (not bins.probed) or N > 1.5 * bins.n_probe
                if not (measure or not need_bwd):
                    break
                torch.cuda.current_stream().synchronize()
                fullest, overflow = int(bins.status_np[0]), int(bins.status_np[1])
                bins.mean_list, bins.fullest = float(int(bins.status_np[3])) / M, fullest
                if measure:
                    bins.probed, bins.n_probe = True, N
                    if overflow or 8 * fullest > bins.slots:
                        bins.grow_to(fullest, 8)
                elif overflow or 2 * fullest > bins.slots:
                    bins.grow_to(fullest, 4)
                if not overflow:
                    break
                if bins.slots < fullest:
                    raise RuntimeError(f"{fullest} Gaussians over one tile exceed the largest bin this image size allows "
                                       f"({bins.limit} slots): call rasterization(..., fused=False)")
                # the lists of this very call were cut: run it again on the larger bins
            slots = flatten_ids.numel() // M
        rv = rec.view(C, N, 16)
        means2d, conics, opac, rgb = rv[:, :, 0:2], rv[:, :, 2:5], rv[:, :, 5], rv[:, :, 6:9]
        depths, radii = rv[:, :, 9], rv[:, :, 10].view(i32)
        if need_bwd:
            ctx.save_for_backward(means, quats, scales, opacities, sh0, shN, viewmats, Ks, backgrounds, rec, vrec, counters,
                                  flatten_ids, render_alphas, last_ids)
            ctx.cfg, ctx.slots, ctx.holder, ctx.bins, ctx.tile_order = cfg, slots, holder, bins, tile_order
        holder["tile_counts"], holder["slots"] = counters[:M], slots
        ctx.mark_non_differentiable(radii, depths, conics, opac, rgb)
        return render_colors, render_alphas, means2d, radii, depths, conics, opac, rgb

    @staticmethod
    def backward(ctx, v_rc, v_ra, v_m2d, *_unused):
        (means, quats, scales, opacities, sh0, shN, viewmats, Ks, backgrounds, rec, vrec, counters, flatten_ids, render_alphas,
         last_ids) = ctx.saved_tensors
        cfg, bins = ctx.cfg, ctx.bins
        W, H = cfg["width"], cfg["height"]
        C, N = viewmats.shape[0], means.shape[0]
        K = sh0.shape[1] if shN is None else 1 + shN.shape[1]
        dev = means.device
        f32 = torch.float32
        if v_rc is None:
            v_rc = torch.zeros(C, H, W, 3, dtype=f32, device=dev)
        if v_ra is None:                     # the photometric loss does not look at alpha: one shared block of zeros
            v_ra = bins.zero_alphas.get((C, H, W))
            if v_ra is None:
                v_ra = bins.zero_alphas[(C, H, W)] = torch.zeros(C, H, W, 1, dtype=f32, device=dev)
        v_rc, v_ra = v_rc.contiguous(), v_ra.contiguous()
        if v_m2d is not None:                # the caller built something on info["means2d"]: its gradient joins the rasteriser's
            vrec.view(C, N, 16)[:, :, 0:2].add_(v_m2d)
        d = _desc(cfg, C, N, K)
        d.bin_capacity = ctx.slots
        # which backward rasteriser: the fused engine's rule (list_policy.pick_raster_impl: much list work, evenly spread, enough tiles),
        # from the lists the last looked-at status word of these bins reported -- no read here
        d.raster_impl = 1 if pick_raster_impl(0, bins.mean_list, bins.fullest, cfg["tile_size"] == 16, bool(cfg["absgrad"]), first=True,
                                              n_tiles=bins.M) == 1 else -1
        if ctx.tile_order is not None:
            d.tile_order = ctx.tile_order.data_ptr()
        d.means, d.quats, d.scales, d.opacities, d.sh0, d.shN = (means.data_ptr(), quats.data_ptr(), scales.data_ptr(),
                                                                   opacities.data_ptr(), sh0.data_ptr(), _p(shN))
        d.viewmats, d.Ks, d.backgrounds = viewmats.data_ptr(), Ks.data_ptr(), _p(backgrounds)
        d.rec, d.vrec, d.counters, d.flatten_ids = rec.data_ptr(), vrec.data_ptr(), counters.data_ptr(), flatten_ids.data_ptr()
        d.render_alphas, d.last_ids = render_alphas.data_ptr(), last_ids.data_ptr()
        d.v_render_colors, d.v_render_alphas = v_rc.data_ptr(), v_ra.data_ptr()
        v_means, v_quats, v_scales = torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales)
        v_opacities = torch.empty_like(opacities)
        v_sh0 = torch.empty(N, 1, 3, dtype=f32, device=dev)
        v_shN = torch.empty(N, K - 1, 3, dtype=f32, device=dev) if K > 1 else None
        g_m2d = torch.empty(C, N, 2, dtype=f32, device=dev)
        g_abs = torch.empty(C, N, 2, dtype=f32, device=dev) if cfg["absgrad"] else None
        d.v_means, d.v_quats, d.v_scales, d.v_opacities = v_means.data_ptr(), v_quats.data_ptr(), v_scales.data_ptr(), v_opacities.data_ptr()
        d.v_sh0, d.v_shN, d.v_means2d, d.v_means2d_abs = v_sh0.data_ptr(), _p(v_shN), g_m2d.data_ptr(), _p(g_abs)
        _lib.call("so_rasterization_bwd", ctypes.byref(d), _lib.stream())
        if cfg["activated"]:                 # one coefficient tensor came in: its gradient goes back as one
            v_sh0 = v_sh0 if K == 1 else torch.cat([v_sh0, v_shN], 1)
            v_shN = None
        # the screen-space gradient the densification strategy accumulates: gsplat leaves it on info["means2d"]
        m2d = ctx.holder.get("means2d")
        m2d = m2d() if m2d is not None else None
        if m2d is not None:
            m2d.grad = g_m2d
            if g_abs is not None:
                m2d.absgrad = g_abs
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[8]:
            v_bg = (v_rc * (1.0 - render_alphas)).sum(dim=(1, 2))
        return v_means, v_quats, v_scales, v_opacities, v_sh0, v_shN, None, None, v_bg, None, None


def fused_rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, *, sh_degree, near_plane, far_plane,
                        radius_clip, eps2d, tile_size, backgrounds, absgrad, antialiased, camera_model, tile_cull, shN=None):
    """The common-shape call: returns (render_colors, render_alphas, per-view tensors..., thunk of the kernels' list length)
    -- `rendering.rasterization` packs the `meta` dict around them.  shN given: `scales` / `opacities` / `colors` are the
    RAW parameters log-scales / opacity logits / sh0 [N,1,3] (see _Rasterization)."""
    activated = shN is None
    cfg = dict(width=int(width), height=int(height), tile_size=int(tile_size), sh_degree=int(sh_degree),
               near_plane=float(near_plane), far_plane=float(far_plane), radius_clip=float(radius_clip), eps2d=float(eps2d),
               absgrad=bool(absgrad), antialiased=bool(antialiased), camera_model=camera_model, tile_cull=bool(tile_cull),
               activated=activated)
    cfg["key"] = tuple(cfg.values())
    holder: dict = {}
    out = _Rasterization.apply(_c(means), _c(quats), _c(scales), _c(opacities), _c(colors), _c(shN), _c(viewmats.detach()),
                               _c(Ks.detach()), _c(backgrounds), cfg, holder)
    holder["means2d"] = weakref.ref(out[2])
    counts, slots = holder.pop("tile_counts"), holder.pop("slots")
    return out + (lambda: counts.clamp(max=slots).sum(),)
