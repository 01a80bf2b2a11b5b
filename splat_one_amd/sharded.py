"""ShardedEngine -- Gaussian-sharded data parallelism over RCCL/xGMI, one view per GPU per step.

This is the reference's own distributed scheme (gsplat `rasterization(distributed=True)`,
/root/reference/utils/gsplat_utils/gsplat_trainer.py:236-238 strided init, :477-494):
every rank owns N/world Gaussians (parameters, Adam state, densification), projects them for ALL
cameras of the step, and the projected Gaussians are exchanged so that rank r can rasterise view r.
Built here from the same C-ABI kernels as the single-GPU engine:

    so_step_inputs        cameras of all ranks, own target image, counters, Adam schedule   (1 launch)
    so_preprocess_fwd     own shard x C=world cameras -> 64-byte records rec_shard[world][cap]
    all_to_all            block c of rec_shard -> rank c          (world-1)/world x 64 B x N per rank
    so_rec_unpack, so_isect_count/fill, so_rasterize_fwd_packed, so_ssim_l1_fused,
    so_rasterize_bwd_packed  on the records of ALL Gaussians for the own view -> vrec_full
    so_shard_flag_put     "my binning pass overflowed" into a spare slot of every block of vrec_full
    all_to_all            block j of vrec_full -> rank j          same volume back
    so_shard_flag_get     any sender overflowed -> the own skip flag (every rank voids the same iteration)
    so_preprocess_bwd     own shard x C=world cameras (sums the cameras) -> parameter gradients
    so_adam_step_dev      own shard

Two exchanges of (world-1)/world x 6.4 MB per rank at 100k Gaussians, against the 2 x 23.6 MB a
replicated-parameter gradient all-reduce moves (the all-reduce path stays available as
`Config.dp_mode = "allreduce"`); xGMI is point-to-point, so at world=2 that is the difference between
~0.1 ms and ~0.35 ms of link time next to a 0.4 ms step.  The loss of a rank enters with weight
1/world (mean over the global batch), exactly like the batch mean of a single-GPU multi-view step.

Shards may differ in length after densification: buffers are sized for `cap` = the largest shard
(agreed with one all-reduce(MAX) whenever a workspace is built) and `cam_stride = cap` lets the
preprocess kernels use them in place; padding rows keep radius 0 and are culled by the binning.

A view whose tile intersections outgrow the buffers of its rank voids the iteration on EVERY rank (the flag rides
on the gradient exchange, so no optimiser applies the incomplete gradients); one step later all ranks find it in
their host-mapped status words, roll their step counters back, agree on a larger capacity and carry on
(`on_overflow = "grow"`, like FusedEngine; "raise" stops instead).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib
from .engine import PARAM_ORDER
from .ops import camera_model_code


def all_to_all_rows(out: Tensor, inp: Tensor, group=None) -> None:
    """Equal-split all-to-all along dim 0 (block j of `inp` goes to rank j).  RCCL directly; the gloo
    backend (CPU tests, single-GPU multi-process tests) stages through host memory."""
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(out, inp, group=group)
    else:
        o = torch.empty(inp.shape, dtype=inp.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)


class ShardedEngine:
    def __init__(self, splats: torch.nn.ParameterDict, optimizers: Dict[str, torch.optim.Optimizer], width: int,
                 height: int, rank: int, world: int, *, sh_degree: int = 3, camera_model: str = "pinhole",
                 near_plane: float = 0.01, far_plane: float = 1e8, radius_clip: float = 0.0, eps2d: float = 0.3,
                 antialiased: bool = False, absgrad: bool = False, ssim_lambda: float = 0.2, opacity_reg: float = 0.0,
                 scale_reg: float = 0.0, tile_size: int = 16, strategy_state: Optional[dict] = None,
                 lr_gamma_means: float = 1.0, isect_capacity: Optional[int] = None, group=None,
                 attr_dtype: str = "f32", tile_cull: bool = True):
        assert dist.is_initialized() and dist.get_world_size(group) == world, "ShardedEngine needs the process group"
        self.splats, self.optimizers = splats, optimizers
        self.W, self.H, self.rank, self.world, self.group = int(width), int(height), int(rank), int(world), group
        self.cfg = dict(sh_degree=sh_degree, camera_model=camera_model, near_plane=near_plane, far_plane=far_plane,
                        radius_clip=radius_clip, eps2d=eps2d, antialiased=antialiased, absgrad=absgrad,
                        ssim_lambda=ssim_lambda, opacity_reg=opacity_reg, scale_reg=scale_reg, tile_size=tile_size)
        assert attr_dtype in ("f32", "f16"), attr_dtype
        self.tile_cull = bool(tile_cull)              # exact tile culling of the binning passes (see FusedEngine)
        self.attr_dtype = attr_dtype                  # "f16": float16 attribute rows of the own shard (see FusedEngine)
        self.strategy_state = strategy_state
        self.lr_gamma_means = lr_gamma_means
        self.device = splats["means"].device
        assert self.device.type == "cuda", "ShardedEngine needs HIP tensors (no CPU path exists)"
        camera_model_code(camera_model, world)       # a name, or the names of the cameras of ALL ranks in rank order
        self._capacity_hint = isect_capacity
        self.steps_done = 0
        self._step_dev = torch.zeros(2 + 4 * _lib.SO_ADAM_MAX_GROUPS, dtype=torch.int32, device=self.device)
        self._status = torch.zeros(4, dtype=torch.int32).pin_memory()    # {n_isects, overflow, seq} of the previous step
        self._seq = 0
        self._status_event = None
        self.on_overflow = "grow"        # "grow": void iteration on every rank, larger buffers, continue;  "raise"
        self.void_steps = 0              # iterations discarded because some rank's binning pass overflowed
        self._build_workspace()

    # ---------------------------------------------------------------------------------------------
    def _build_workspace(self) -> None:
        dev, W, H, n = self.device, self.W, self.H, self.world
        self._ptr_cache, self._adam_cache = {}, None         # everything below is reallocated
        N = self.splats["means"].shape[0]
        K = 1 + self.splats["shN"].shape[1]
        sizes = torch.tensor([N, -N], dtype=torch.int64, device=dev)         # max and (negated) min in one collective
        dist.all_reduce(sizes, op=dist.ReduceOp.MAX, group=self.group)
        cap = int(sizes[0].item())
        tot = torch.tensor([N], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        self.N, self.K, self.cap, self.N_total = N, K, cap, int(tot.item())
        Nf = self.Nf = n * cap
        ts = self.cfg["tile_size"]
        tw, th = math.ceil(W / ts), math.ceil(H / ts)
        self.M = M = tw * th
        icap_t = torch.tensor([self._capacity_hint or max(1 << 20, 8 * Nf)], dtype=torch.int64, device=dev)
        dist.all_reduce(icap_t, op=dist.ReduceOp.MAX, group=self.group)     # one capacity on every rank
        icap = int(icap_t.item())
        self.capacity = int(icap)
        self._probe_capacity = self._capacity_hint is None
        f32, i32 = torch.float32, torch.int32
        e = lambda *shape, dtype=f32: torch.empty(*shape, dtype=dtype, device=dev)
        z = lambda *shape, dtype=f32: torch.zeros(*shape, dtype=dtype, device=dev)
        w = self.ws = {}
        w["viewmats"], w["Ks"] = e(n, 4, 4), e(n, 3, 3)
        # own shard x all cameras (camera-major with stride cap; rows >= N are never written: radius 0)
        w["radii"], w["tiles_per_gauss"] = z(n, cap, dtype=i32), z(n, cap, dtype=i32)
        w["means2d"], w["depths"], w["conics"] = z(n, cap, 2), z(n, cap), z(n, cap, 3)
        w["opacities"], w["colors"] = z(n, cap), z(n, cap, 3)
        w["rec_shard"], w["vrec_shard"] = z(Nf, 16), e(Nf, 16)
        # all Gaussians x own camera
        w["rec_full"], w["vrec_full"] = e(Nf, 16), e(Nf, 16)
        w["means2d_full"], w["radii_full"], w["depths_full"] = e(Nf, 2), e(Nf, dtype=i32), e(Nf)
        w["tiles_full"] = e(Nf, dtype=i32)
        w["counters"] = z(2 * M + 9, dtype=i32)        # ... | loss sums (2) | loss, l1, ssimloss (3) | loss-kernel ticket (1)
        w["isect_offsets"] = e(th, tw, dtype=i32)
        w["key_buf"] = z(icap, dtype=torch.int64)
        w["flatten_ids"] = e(icap, dtype=i32)
        w["render_colors"], w["render_alphas"] = e(1, H, W, 3), e(1, H, W, 1)
        w["last_ids"] = e(1, H, W, dtype=i32)
        w["loss_sums"] = w["counters"][2 * M + 3:2 * M + 9].view(torch.float32)
        w["v_render_colors"] = e(1, H, W, 3)
        w["zero_v_alphas"] = z(1, H, W)
        w["pixels"] = e(1, H, W, 3)
        pad = lambda k: (k + 63) // 64 * 64
        total = sum(pad(self.splats[k].numel()) for k in PARAM_ORDER)
        w["grads_flat"] = z(total)
        w["grads"], off = {}, 0
        for k in PARAM_ORDER:
            m = self.splats[k].numel()
            w["grads"][k] = w["grads_flat"][off:off + m].view_as(self.splats[k])
            off += pad(m)
        for k in PARAM_ORDER:
            self.splats[k].grad = w["grads"][k]
        if self.attr_dtype == "f16":
            self.attr_stride = int(_lib.load().so_attr_rec_stride(K))
            w["arec"] = e(max(N, 1) * self.attr_stride // 4)
            if N > 0:
                sp, p = self.splats, _lib.ptr
                _lib.call("so_attr_pack_f16", N, K, p(sp["scales"].data), p(sp["quats"].data), p(sp["sh0"].data),
                          p(sp["shN"].data), p(w["arec"]), _lib.stream())
        if self.strategy_state is not None:
            for k in ("grad2d", "count"):
                if self.strategy_state.get(k) is None or self.strategy_state[k].shape[0] != N:
                    self.strategy_state[k] = torch.zeros(N, device=dev)

    def rebuild(self) -> None:
        """Call on EVERY rank after a densification step (the shard sizes are re-agreed collectively)."""
        self._build_workspace()

    def _P(self, key: str) -> int:
        """Cached device pointer: "w.<workspace tensor>", "s.<parameter>", "g.<gradient>", "c.<slot of the counters>"."""
        v = self._ptr_cache.get(key)
        if v is None:
            kind, _, name = key.partition(".")
            M = self.M
            if kind == "w":
                t = self.ws[name]
            elif kind == "s":
                t = self.splats[name].data
            elif kind == "g":
                t = self.ws["grads"][name]
            else:
                t = {"cursor": self.ws["counters"][M:], "n_isects": self.ws["counters"][2 * M + 1:],
                     "overflow": self.ws["counters"][2 * M + 2:], "loss_out": self.ws["loss_sums"][2:],
                     "loss_ticket": self.ws["loss_sums"][5:]}[name]
            v = self._ptr_cache[key] = _lib.ptr(t)
        return v

    def set_sh_degree(self, deg: int) -> None:
        self.cfg["sh_degree"] = deg

    # same schedule bookkeeping as FusedEngine
    def _adam_args(self):
        if self._adam_cache is not None:        # parameter / moment tensors only change with rebuild()
            return self._adam_cache
        items = []
        for k in PARAM_ORDER:
            opt = self.optimizers[k]
            grp = opt.param_groups[0]
            prm = self.splats[k]
            st = opt.state[prm]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(prm)
                st["exp_avg_sq"] = torch.zeros_like(prm)
            items.append((prm, st, grp))
        betas, eps = items[0][2]["betas"], items[0][2]["eps"]
        assert all(i[2]["betas"] == betas and i[2]["eps"] == eps for i in items), "one (betas, eps) class expected"
        n = len(items)
        arr = (_lib.AdamGroup * n)()
        lr0 = (ctypes.c_float * n)()
        gam = (ctypes.c_float * n)()
        for i, (k, (prm, st, grp)) in enumerate(zip(PARAM_ORDER, items)):
            arr[i] = _lib.AdamGroup(_lib.ptr(prm.data), _lib.ptr(self.ws["grads"][k]), _lib.ptr(st["exp_avg"]),
                                    _lib.ptr(st["exp_avg_sq"]), 0, prm.numel(), max(1, prm[0].numel() if len(prm) else 1), 0.0, 0.0)
            g = self.lr_gamma_means if k == "means" else 1.0
            lr0[i] = grp["lr"] / (g ** self.steps_done)
            gam[i] = g
        self._adam_cache = (n, arr, lr0, gam, betas, eps)
        return self._adam_cache

    # ---------------------------------------------------------------------------------------------
    def fwd_bwd(self, camtoworlds: Tensor, Ks: Tensor, pixels: Tensor, schedule: bool = False) -> None:
        """camtoworlds[world,4,4] / Ks[world,3,3]: the cameras of ALL ranks for this step (rank r renders
        camera r); pixels[1,H,W,3]: the own target image.  Leaves the shard's gradients in `.grad`."""
        self._check_previous()       # may rebuild the workspace (buffers grown after an overflow): before any of it is used
        n, N, cap, Nf, M, W, H = self.world, self.N, self.cap, self.Nf, self.M, self.W, self.H
        assert camtoworlds.shape == (n, 4, 4) and Ks.shape == (n, 3, 3), (camtoworlds.shape, Ks.shape)
        assert pixels.shape == (1, H, W, 3), pixels.shape
        w, s, c, p, st, dev = self.ws, self.splats, self.cfg, _lib.ptr, _lib.stream(), self.device
        P = self._P          # device pointers of the static buffers, looked up once per workspace (the step is a
                             # sequence of ~20 C-ABI calls from Python: their argument marshalling is its host cost)
        c2w = camtoworlds.detach().to(device=dev, dtype=torch.float32).contiguous()
        Ksd = Ks.detach().to(device=dev, dtype=torch.float32).contiguous()
        if pixels.is_cuda and pixels.dtype == torch.float32 and pixels.is_contiguous():
            px = pixels.detach()
        else:
            w["pixels"].copy_(pixels, non_blocking=True)
            px = w["pixels"]
        ng, lr0, gam, betas = 0, None, None, (0.0, 0.0)
        if schedule:
            ng, _arr, lr0, gam, betas, _eps = self._adam_args()
        self._seq = (self._seq + 1) & 0x3FFFFFFF
        _lib.call("so_step_inputs", n, p(c2w), p(Ksd), P("w.viewmats"), P("w.Ks"), 0, 0, P("w.counters"), 2 * M + 5, ng,
                  lr0, gam, float(betas[0]), float(betas[1]), p(self._step_dev), self._status.data_ptr(), 2 * M + 1,
                  self._seq, 0, 0, 0, 0, 0, st)
        self._status_event = torch.cuda.Event()
        self._status_event.record()
        ts = c["tile_size"]
        tw, th = math.ceil(W / ts), math.ceil(H / ts)
        cam = camera_model_code(c["camera_model"], n)
        own_model = c["camera_model"] if isinstance(c["camera_model"], str) else list(c["camera_model"])[self.rank]
        # this rank's image is periodic in x when its camera is a panorama (include/splat_one_amd.h SO_TILE_WRAP_ALL)
        tsw = ts | ((1 << 24) if own_model == "spherical" and W % ts == 0 else 0)
        f16 = self.attr_dtype == "f16"
        if N > 0 and f16:
            _lib.call("so_preprocess_fwd_f16", n, N, self.K, c["sh_degree"], P("s.means"), P("s.opacities"),
                      P("w.arec"), P("w.viewmats"), P("w.Ks"), W, H, c["eps2d"], c["near_plane"], c["far_plane"],
                      c["radius_clip"], cam, int(c["antialiased"]), ts, P("w.radii"), P("w.means2d"), P("w.depths"),
                      P("w.conics"), P("w.opacities"), P("w.colors"), P("w.tiles_per_gauss"), 0, P("w.rec_shard"), 0, cap, 0, 0, 0, 0, 0, st)
        elif N > 0:
            _lib.call("so_preprocess_fwd", n, N, self.K, c["sh_degree"], P("s.means"), P("s.scales"),
                      P("s.quats"), P("s.opacities"), P("s.sh0"), P("s.shN"), P("w.viewmats"),
                      P("w.Ks"), W, H, c["eps2d"], c["near_plane"], c["far_plane"], c["radius_clip"], cam,
                      int(c["antialiased"]), ts, P("w.radii"), P("w.means2d"), P("w.depths"), P("w.conics"),
                      P("w.opacities"), P("w.colors"), P("w.tiles_per_gauss"), 0, P("w.rec_shard"), 0, cap, 0, 0, 0, 0, 0, st)
        all_to_all_rows(w["rec_full"], w["rec_shard"], self.group)
        _lib.call("so_rec_unpack", Nf, P("w.rec_full"), P("w.means2d_full"), P("w.radii_full"), P("w.depths_full"),
                  P("w.vrec_full"), st)
        _lib.call("so_isect_count", 1, Nf, P("w.means2d_full"), P("w.radii_full"), tsw, tw, th, P("w.tiles_full"),
                  P("w.counters"), P("w.isect_offsets"), P("c.n_isects"), P("w.rec_full") if self.tile_cull else 0, st)
        if self._probe_capacity:     # once per workspace: the largest intersection count over the ranks decides the buffers
            self._probe_capacity = False
            cnt = w["counters"][2 * M + 1:2 * M + 2].to(torch.int64)
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX, group=self.group)
            if 1.25 * int(cnt.item()) > self.capacity:
                self._capacity_hint = 2 * int(cnt.item()) + 4096
                self._status_event = None
                self._step_dev[0] = self.steps_done
                self._build_workspace()
                return self.fwd_bwd(camtoworlds, Ks, pixels, schedule)
        _lib.call("so_isect_fill", 1, Nf, P("w.means2d_full"), P("w.radii_full"), P("w.depths_full"), tsw, tw, th,
                  P("w.isect_offsets"), P("c.n_isects"), P("c.cursor"), self.capacity, P("w.key_buf"), P("w.flatten_ids"), 0,
                  P("c.overflow"), 0, P("w.rec_full") if self.tile_cull else 0, st)
        _lib.call("so_rasterize_fwd_packed", 1, Nf, W, H, tsw, P("w.rec_full"), 0, P("w.isect_offsets"), P("w.flatten_ids"),
                  P("c.n_isects"), self.capacity, P("w.render_colors"), P("w.render_alphas"), P("w.last_ids"), st)
        lam = float(c["ssim_lambda"])
        n_l1 = float(H * W * 3)
        n_ss = float((H - 10) * (W - 10) * 3)
        # weight 1/world: the step's loss is the mean over the global batch of views
        _lib.call("so_ssim_l1_fused", 1, H, W, 3, P("w.render_colors"), p(px), 1, (1.0 - lam) / n_l1 / n, -lam / n_ss / n, 0,
                  P("w.loss_sums"), P("w.v_render_colors"), P("c.loss_out"), P("c.loss_ticket"), lam / n, 0, st)
        _lib.call("so_rasterize_bwd_packed", 1, Nf, W, H, tsw, P("w.rec_full"), 0, P("w.isect_offsets"), P("w.flatten_ids"),
                  P("c.n_isects"), self.capacity, P("w.render_alphas"), P("w.last_ids"), P("w.v_render_colors"), P("w.zero_v_alphas"),
                  P("w.vrec_full"), int(c["absgrad"]), st)
        _lib.call("so_shard_flag_put", n, cap, P("c.overflow"), P("w.vrec_full"), st)
        all_to_all_rows(w["vrec_shard"], w["vrec_full"], self.group)
        _lib.call("so_shard_flag_get", n, cap, P("w.vrec_shard"), P("c.overflow"), st)
        if N > 0:
            sst = self.strategy_state
            # the regularisers are means over ALL Gaussians: rescale the kernel's 1/N to 1/N_total
            scale = float(N) / float(max(self.N_total, 1))
            if f16:
                _lib.call("so_preprocess_bwd_f16", n, N, self.K, c["sh_degree"], P("s.means"), P("s.opacities"),
                          P("w.arec"), P("w.viewmats"), P("w.Ks"), W, H, c["eps2d"], cam, int(c["antialiased"]),
                          P("w.radii"), P("w.opacities"), P("w.colors"), c["opacity_reg"] * scale, c["scale_reg"] * scale,
                          P("g.means"), P("g.scales"), P("g.quats"), P("g.opacities"), P("g.sh0"), P("g.shN"),
                          p(sst["grad2d"]) if sst is not None else 0, p(sst["count"]) if sst is not None else 0,
                          P("w.vrec_shard"), int(c["absgrad"]), cap, P("c.overflow"), 0, st)
            else:
                _lib.call("so_preprocess_bwd", n, N, self.K, c["sh_degree"], P("s.means"), P("s.scales"),
                          P("s.quats"), P("s.opacities"), P("s.sh0"), P("s.shN"), P("w.viewmats"),
                          P("w.Ks"), W, H, c["eps2d"], cam, int(c["antialiased"]), P("w.radii"), P("w.opacities"),
                          P("w.colors"), 0, 0, 0, 0, 0, 0, c["opacity_reg"] * scale, c["scale_reg"] * scale, P("g.means"),
                          P("g.scales"), P("g.quats"), P("g.opacities"), P("g.sh0"), P("g.shN"),
                          p(sst["grad2d"]) if sst is not None else 0, p(sst["count"]) if sst is not None else 0,
                          P("w.vrec_shard"), int(c["absgrad"]), cap, P("c.overflow"), 0, st)
        self._keep = (c2w, Ksd, px)
        self._sched_staged = bool(schedule)

    def _check_previous(self) -> None:
        """One step late, without a device-wide sync: did the binning pass of ANY rank overflow its buffers in the
        iteration before the last one?  The flag every rank reads here is the OR over the ranks
        (so_shard_flag_put / _get around the gradient exchange), so all ranks take this branch on the same step:
        the kernels stayed in bounds, every optimiser skipped; here the host-side step counters are rolled back and
        the buffers grow to a capacity agreed in `_build_workspace` (a collective: every rank is in it)."""
        ev, self._status_event = self._status_event, None
        if ev is None:
            return
        ev.synchronize()
        n_prev, ov_prev, seq = (int(v) for v in self._status[:3])
        if seq != self._seq or not ov_prev:
            return
        torch.cuda.synchronize()
        cnt = self.ws["counters"]
        n_last, ov_last = int(cnt[2 * self.M + 1].item()), int(cnt[2 * self.M + 2].item())
        need = max(n_prev, n_last)
        if self.on_overflow == "raise":
            raise RuntimeError(f"rank {self.rank}: tile-intersection buffers overflowed on some rank (own count {need}, "
                               f"capacity {self.capacity}); the affected iterations were skipped on every rank -- "
                               "raise Config.isect_capacity")
        void = 1 + (1 if ov_last else 0)
        self.void_steps += void
        for _ in range(void):                        # undo optimize()'s bookkeeping for iterations that never happened
            self.steps_done -= 1
            for k in PARAM_ORDER:
                self.optimizers[k].state[self.splats[k]]["step"] -= 1
            self.optimizers["means"].param_groups[0]["lr"] /= self.lr_gamma_means
        self._step_dev[0] = self.steps_done
        import warnings
        warnings.warn(f"splat_one_amd: rank {self.rank}: {void} training iteration(s) skipped on every rank -- the tile "
                      f"intersections of some view exceeded the buffer capacity {self.capacity}; buffers enlarged",
                      RuntimeWarning)
        self._capacity_hint = max(2 * self.capacity, int(1.5 * need) + 4096)
        self._probe_capacity = False
        self._build_workspace()

    def optimize(self) -> None:
        n, arr, lr0, gam, betas, eps = self._adam_args()
        sched = getattr(self, "_sched_staged", False)
        self._sched_staged = False
        shadow = None
        repack = self.attr_dtype == "f16" and self.N > 0 and getattr(self, "f16_repack", True)    # see FusedEngine._launch_optimize
        if self.attr_dtype == "f16" and self.N > 0 and not repack:
            where = {"scales": 8, "quats": 0, "sh0": 16, "shN": 22}
            shadow = _lib.AttrShadow(_lib.ptr(self.ws["arec"]), self.attr_stride,
                                     (ctypes.c_int32 * _lib.SO_ADAM_MAX_GROUPS)(*(
                                         [where.get(k, -1) if self.splats[k].numel() else -1 for k in PARAM_ORDER]
                                         + [-1] * (_lib.SO_ADAM_MAX_GROUPS - len(PARAM_ORDER)))))
        _lib.call("so_adam_step_dev_shadow", n, arr, lr0, gam, float(betas[0]), float(betas[1]), float(eps),
                  _lib.ptr(self._step_dev), 0, int(sched), self._P("c.overflow"), 0,
                  ctypes.byref(shadow) if shadow is not None else None, _lib.stream())
        if repack:
            sp, p = self.splats, _lib.ptr
            _lib.call("so_attr_pack_f16", self.N, self.K, p(sp["scales"].data), p(sp["quats"].data), p(sp["sh0"].data),
                      p(sp["shN"].data), self._P("w.arec"), _lib.stream())
        self.steps_done += 1
        for k in PARAM_ORDER:
            self.optimizers[k].state[self.splats[k]]["step"] += 1
        self.optimizers["means"].param_groups[0]["lr"] *= self.lr_gamma_means

    def step(self, camtoworlds: Tensor, Ks: Tensor, pixels: Tensor) -> None:
        self.fwd_bwd(camtoworlds, Ks, pixels, schedule=True)
        self.optimize()

    def loss(self) -> Tensor:
        """(loss/world, l1, ssimloss) of the own view for the last step (device floats; the first entry carries
        the 1/world weight it entered the global mean with)."""
        return self.ws["loss_sums"][2:5]

    def stats(self) -> dict:
        c = self.ws["counters"]
        return {"n_isects": int(c[2 * self.M + 1].item()), "overflow": int(c[2 * self.M + 2].item()),
                "visible": int((self.ws["radii_full"] > 0).sum().item())}
