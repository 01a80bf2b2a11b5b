"""Training engine of the path -- the slice of the reference's `Runner`
(/root/reference/utils/gsplat_utils/gsplat_trainer.py) that is the hot loop:

    Config                        :62-201   (fields that reach the rasteriser / optimiser / strategy)
    create_splats_with_optimizers :204-281
    Runner.rasterize_splats       :446-497
    Runner.train (one iteration)  :551-763  pre-backward hook -> loss -> backward -> optimisers ->
                                            LR schedule -> post-backward (densification)

Same names, argument meaning and step order, so `app/gsplat_manager.py` style callers
(`Runner(local_rank, world_rank, world_size, cfg)`, `.train()`, `.rasterize_splats(...)`) read the
same.  Out of scope here: dataset parsers, viewer, TensorBoard, eval,
compression, pose/appearance/bilateral-grid modules.

Multi-GPU: view-sharded data parallelism (splat_one_amd.distributed), not the reference's
Gaussian sharding; the random init is therefore NOT strided over ranks (every rank holds all
Gaussians) unless `shard_gaussians=True` reproduces `points[world_rank::world_size]` (:236-238).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import Tensor

from . import distributed as sdist
from .losses import photometric_loss
from .optimizers import FusedAdam, SelectiveAdam, step_all
from .rendering import rasterization
from .scene import knn, rgb_to_sh, set_random_seed
from .strategy import DefaultStrategy, MCMCStrategy


@dataclass
class Config:
    # names and defaults follow gsplat_trainer.py:62-182
    batch_size: int = 1
    steps_scaler: float = 1.0
    max_steps: int = 30_000
    init_type: str = "random"
    init_num_pts: int = 100_000
    init_extent: float = 3.0
    sh_degree: int = 3
    sh_degree_interval: int = 1000
    init_opa: float = 0.1
    init_scale: float = 1.0
    ssim_lambda: float = 0.2
    near_plane: float = 0.01
    far_plane: float = 1e8
    strategy: Union[DefaultStrategy, MCMCStrategy] = field(default_factory=DefaultStrategy)
    packed: bool = False
    sparse_grad: bool = False
    visible_adam: bool = False
    antialiased: bool = False
    random_bkgd: bool = False
    opacity_reg: float = 0.0
    scale_reg: float = 0.0
    global_scale: float = 1.0
    camera_model: str = "pinhole"   # reference default "spherical" is fork-only (no specification)
    # extensions of this build
    isect_capacity: Optional[int] = None   # preallocated intersections -> no host sync in the step
    fused: bool = False                    # FusedEngine: whole step in two C-ABI calls, hipGraph replay
    shN_init_std: float = 0.0              # >0: noise instead of zeros in the higher SH bands (bench scenes)

    def adjust_steps(self, factor: float):
        """gsplat_trainer.py:184-201"""
        self.max_steps = int(self.max_steps * factor)
        self.sh_degree_interval = int(self.sh_degree_interval * factor)
        s = self.strategy
        if isinstance(s, DefaultStrategy):
            s.refine_start_iter = int(s.refine_start_iter * factor)
            s.refine_stop_iter = int(s.refine_stop_iter * factor)
            s.reset_every = int(s.reset_every * factor)
            s.refine_every = int(s.refine_every * factor)
        elif isinstance(s, MCMCStrategy):
            s.refine_start_iter = int(s.refine_start_iter * factor)
            s.refine_stop_iter = int(s.refine_stop_iter * factor)
            s.refine_every = int(s.refine_every * factor)


PARAM_LRS = (("means", 1.6e-4), ("scales", 5e-3), ("quats", 1e-3), ("opacities", 5e-2),
             ("sh0", 2.5e-3), ("shN", 2.5e-3 / 20))


def adam_hyperparameters(lr: float, batch_size: int, world_size: int) -> Tuple[float, float, Tuple[float, float]]:
    """The batch-size scaling rule of gsplat_trainer.py:266-278 -> (lr, eps, betas)."""
    BS = batch_size * world_size
    return lr * math.sqrt(BS), 1e-15 / math.sqrt(BS), (1 - BS * (1 - 0.9), 1 - BS * (1 - 0.999))


def create_splats_with_optimizers(
    points: Optional[Tensor] = None, rgbs: Optional[Tensor] = None, init_type: str = "random",
    init_num_pts: int = 100_000, init_extent: float = 3.0, init_opacity: float = 0.1,
    init_scale: float = 1.0, scene_scale: float = 1.0, sh_degree: int = 3, sparse_grad: bool = False,
    visible_adam: bool = False, batch_size: int = 1, device: str = "cuda", world_rank: int = 0,
    world_size: int = 1, shard_gaussians: bool = False, shN_init_std: float = 0.0,
) -> Tuple[torch.nn.ParameterDict, Dict[str, torch.optim.Optimizer]]:
    """Restates gsplat_trainer.py:204-281 (init_type "sfm" takes `points`/`rgbs` in place of the parser)."""
    if init_type == "sfm":
        assert points is not None and rgbs is not None, "sfm init needs points [N,3] and rgbs [N,3] in 0..1"
        points, rgbs = points.float(), rgbs.float()
    elif init_type == "random":
        points = init_extent * scene_scale * (torch.rand((init_num_pts, 3)) * 2 - 1)
        rgbs = torch.rand((init_num_pts, 3))
    else:
        raise ValueError("Please specify a correct init_type: sfm or random")
    dist2_avg = (knn(points, 4)[:, 1:] ** 2).mean(dim=-1)
    dist_avg = torch.sqrt(dist2_avg)
    scales = torch.log(dist_avg * init_scale).unsqueeze(-1).repeat(1, 3)
    if shard_gaussians:
        points, rgbs, scales = (t[world_rank::world_size] for t in (points, rgbs, scales))
    N = points.shape[0]
    quats = torch.rand((N, 4))
    opacities = torch.logit(torch.full((N,), init_opacity))
    colors = torch.zeros((N, (sh_degree + 1) ** 2, 3))
    colors[:, 0, :] = rgb_to_sh(rgbs)
    if shN_init_std > 0:
        g = torch.Generator().manual_seed(43)
        colors[:, 1:, :] = torch.randn(colors[:, 1:, :].shape, generator=g) * shN_init_std
    values = {"means": points, "scales": scales, "quats": quats, "opacities": opacities,
              "sh0": colors[:, :1, :].contiguous(), "shN": colors[:, 1:, :].contiguous()}
    lrs = dict(PARAM_LRS)
    lrs["means"] = lrs["means"] * scene_scale
    splats = torch.nn.ParameterDict({n: torch.nn.Parameter(v) for n, v in values.items()}).to(device)
    assert not sparse_grad, "sparse_grad needs packed mode, which is not implemented yet"
    opt_cls = SelectiveAdam if visible_adam else FusedAdam
    optimizers = {}
    for name in values:
        lr, eps, betas = adam_hyperparameters(lrs[name], batch_size, world_size)
        optimizers[name] = opt_cls([{"params": splats[name], "lr": lr, "name": name}], eps=eps, betas=betas)
    return splats, optimizers


class Runner:
    """Engine for training (hot path only).  `views` replaces the reference's dataset/parser:
    a list of dicts {"K":[3,3], "camtoworld":[4,4], "image":[H,W,3] in 0..255} like
    `Dataset.__getitem__` (utils/datasets/opensfm.py:341-389)."""

    def __init__(self, local_rank: int, world_rank: int, world_size: int, cfg: Config,
                 views: Optional[List[Dict[str, Tensor]]] = None, scene_scale: float = 1.0,
                 points: Optional[Tensor] = None, rgbs: Optional[Tensor] = None):
        # reference: set_random_seed(42 + local_rank) (:290); here Gaussians are replicated, so every
        # rank must draw the same stream
        set_random_seed(42)
        self.cfg = cfg
        self.world_rank, self.local_rank, self.world_size = world_rank, local_rank, world_size
        self.device = f"cuda:{local_rank}"
        self.views = views or []
        self.scene_scale = scene_scale * 1.1 * cfg.global_scale          # gsplat_trainer.py:322
        self.splats, self.optimizers = create_splats_with_optimizers(
            points, rgbs, init_type=cfg.init_type, init_num_pts=cfg.init_num_pts, init_extent=cfg.init_extent,
            init_opacity=cfg.init_opa, init_scale=cfg.init_scale, scene_scale=self.scene_scale,
            sh_degree=cfg.sh_degree, sparse_grad=cfg.sparse_grad, visible_adam=cfg.visible_adam,
            batch_size=cfg.batch_size, device=self.device, world_rank=world_rank, world_size=world_size,
            shN_init_std=cfg.shN_init_std)
        self.cfg.strategy.check_sanity(self.splats, self.optimizers)
        if isinstance(self.cfg.strategy, MCMCStrategy):
            self.strategy_state = self.cfg.strategy.initialize_state()           # gsplat_trainer.py:351-352
        else:
            self.strategy_state = self.cfg.strategy.initialize_state(scene_scale=self.scene_scale)
        self.means_lr0 = self.optimizers["means"].param_groups[0]["lr"]
        self.lr_gamma = 0.01 ** (1.0 / cfg.max_steps)                     # ExponentialLR, :512-516
        self.step = 0
        self.stop_training = False
        self._workspace: dict = {}
        self._reducer = sdist.GradientReducer()
        self._split_gen = torch.Generator(device=self.device)
        self._split_gen.manual_seed(1234)                                 # same on every rank
        self.last_info: Optional[dict] = None

    # ------------------------------------------------------------------------------ :446-497
    def rasterize_splats(self, camtoworlds: Tensor, Ks: Tensor, width: int, height: int,
                         masks: Optional[Tensor] = None, camera_model: Optional[str] = None,
                         **kwargs) -> Tuple[Tensor, Tensor, Dict]:
        means = self.splats["means"]
        quats = self.splats["quats"]
        scales = torch.exp(self.splats["scales"])
        opacities = torch.sigmoid(self.splats["opacities"])
        if camera_model is None:
            camera_model = self.cfg.camera_model
        kwargs.pop("image_ids", None)
        colors = torch.cat([self.splats["sh0"], self.splats["shN"]], 1)
        rasterize_mode = "antialiased" if self.cfg.antialiased else "classic"
        render_colors, render_alphas, info = rasterization(
            means=means, quats=quats, scales=scales, opacities=opacities, colors=colors,
            viewmats=torch.linalg.inv(camtoworlds), Ks=Ks, width=width, height=height,
            packed=self.cfg.packed,
            absgrad=(self.cfg.strategy.absgrad if isinstance(self.cfg.strategy, DefaultStrategy) else False),
            sparse_grad=self.cfg.sparse_grad, rasterize_mode=rasterize_mode, distributed=False,
            camera_model=camera_model, isect_capacity=self.cfg.isect_capacity, workspace=self._workspace,
            **kwargs)
        if masks is not None:
            render_colors[~masks] = 0
        return render_colors, render_alphas, info

    # ------------------------------------------------------------------------------ :551-763
    # ------------------------------------------------------------------------------ fused fast path
    def _fused_ok(self, masks) -> bool:
        c = self.cfg
        if not (c.fused and masks is None and not c.random_bkgd and not c.visible_adam and not c.packed):
            return False
        if isinstance(c.strategy, DefaultStrategy):
            return c.strategy.refine_scale2d_stop_iter == 0
        return isinstance(c.strategy, MCMCStrategy)

    def _train_step_fused(self, camtoworlds: Tensor, Ks: Tensor, pixels: Tensor) -> Tensor:
        from .engine import FusedEngine
        cfg, step, s = self.cfg, self.step, self.cfg.strategy
        B, H, W = pixels.shape[0], pixels.shape[1], pixels.shape[2]
        eng = getattr(self, "_engine", None)
        if eng is None or (eng.C, eng.H, eng.W) != (B, H, W):
            eng = self._engine = FusedEngine(
                self.splats, self.optimizers, W, H, B, sh_degree=0, camera_model=cfg.camera_model,
                near_plane=cfg.near_plane, far_plane=cfg.far_plane, antialiased=cfg.antialiased,
                absgrad=getattr(s, "absgrad", False),
                ssim_lambda=cfg.ssim_lambda, opacity_reg=cfg.opacity_reg, scale_reg=cfg.scale_reg,
                strategy_state=(self.strategy_state if isinstance(s, DefaultStrategy) else None),
                lr_gamma_means=self.lr_gamma,
                isect_capacity=cfg.isect_capacity, use_graph=True,
                raster_impl=getattr(self, "raster_impl", 0))
            eng.steps_done = step
            eng._step_dev[0] = step
        eng.set_sh_degree(min(step // cfg.sh_degree_interval, cfg.sh_degree))
        # densification statistics are accumulated inside the backward kernel while refinement is active
        stats_on = isinstance(s, DefaultStrategy) and step < s.refine_stop_iter
        if stats_on != (eng.strategy_state is not None):
            eng.strategy_state = self.strategy_state if stats_on else None
            eng._graph = None
            eng._graph_fb = eng._graph_opt = None
        eng.set_views(camtoworlds, Ks, pixels)
        if self.world_size == 1:
            eng.step()
        else:
            eng.fwd_bwd()
            sdist.all_reduce_mean_(eng.ws["grads_flat"])     # ONE collective on the flat gradient SoA
            eng.optimize()
        if isinstance(s, MCMCStrategy):
            # lr = the means learning rate after this step's scheduler.step() (gsplat_trainer.py:753-761)
            n_before = len(self.splats["means"])
            n_rel, n_new = s.step_post_backward(params=self.splats, optimizers=self.optimizers,
                                                state=self.strategy_state, step=step, info={},
                                                lr=self.optimizers["means"].param_groups[0]["lr"],
                                                generator=self._split_gen)
            if n_rel or n_new:
                eng.rebuild()
            self.last_info = {"radii": eng.ws["radii"], "n_isects": eng.ws["counters"][2 * eng.M + 1:2 * eng.M + 2],
                              "flatten_ids": eng.ws["flatten_ids"], "means2d": eng.ws["means2d"]}
            self.step += 1
            return eng.loss()[0]
        refine_now = (step < s.refine_stop_iter and step > s.refine_start_iter and step % s.refine_every == 0
                      and step % s.reset_every >= s.pause_refine_after_reset)
        reset_now = step < s.refine_stop_iter and step % s.reset_every == 0 and step > 0
        if refine_now or reset_now:
            n_before = len(self.splats["means"])
            if refine_now:
                if self.world_size > 1:
                    sdist.all_reduce_strategy_state(self.strategy_state)
                n_dupli, n_split = s._grow_gs(self.splats, self.optimizers, self.strategy_state, step, self._split_gen)
                n_prune = s._prune_gs(self.splats, self.optimizers, self.strategy_state, step)
                if s.verbose:
                    print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split, {n_prune} GSs pruned. "
                          f"Now having {len(self.splats['means'])} GSs.")
                self.strategy_state["grad2d"].zero_()
                self.strategy_state["count"].zero_()
            if reset_now:
                from .strategy import reset_opa
                reset_opa(params=self.splats, optimizers=self.optimizers, state=self.strategy_state,
                          value=s.prune_opa * 2.0)
            eng.rebuild()
        self.last_info = {"radii": eng.ws["radii"], "n_isects": eng.ws["counters"][2 * eng.M + 1:2 * eng.M + 2],
                          "flatten_ids": eng.ws["flatten_ids"], "means2d": eng.ws["means2d"]}
        self.step += 1
        return eng.loss()[0]

    def train_step(self, camtoworlds: Tensor, Ks: Tensor, pixels: Tensor, masks: Optional[Tensor] = None) -> Tensor:
        """One iteration on an already-on-device batch (camtoworlds[B,4,4], Ks[B,3,3],
        pixels[B,H,W,3] in 0..1).  Returns the loss tensor (no host sync; on the fused path it is a
        view of the engine's static loss buffer, valid until the next step -- clone it to keep it)."""
        if self._fused_ok(masks):
            return self._train_step_fused(camtoworlds, Ks, pixels)
        cfg, step = self.cfg, self.step
        height, width = pixels.shape[1:3]
        sh_degree_to_use = min(step // cfg.sh_degree_interval, cfg.sh_degree)
        renders, alphas, info = self.rasterize_splats(
            camtoworlds=camtoworlds, Ks=Ks, width=width, height=height, sh_degree=sh_degree_to_use,
            near_plane=cfg.near_plane, far_plane=cfg.far_plane, render_mode="RGB", masks=masks)
        colors = renders[..., 0:3]
        if cfg.random_bkgd:
            bkgd = torch.rand(1, 3, device=colors.device)
            colors = colors + bkgd * (1.0 - alphas)
        cfg.strategy.step_pre_backward(params=self.splats, optimizers=self.optimizers,
                                       state=self.strategy_state, step=step, info=info)
        loss, _l1, _ssim = photometric_loss(colors, pixels, cfg.ssim_lambda)
        if cfg.opacity_reg > 0.0:
            loss = loss + cfg.opacity_reg * torch.abs(torch.sigmoid(self.splats["opacities"])).mean()
        if cfg.scale_reg > 0.0:
            loss = loss + cfg.scale_reg * torch.abs(torch.exp(self.splats["scales"])).mean()
        loss.backward()
        # view-sharded data parallelism: one all-reduce of the flattened gradient SoA
        if self.world_size > 1:
            self._reducer.reduce(self.splats.values())
        # optimisers (one fused launch) + zero_grad(set_to_none=True)
        vis = None
        if cfg.visible_adam:
            vis = (info["radii"] > 0).any(0)
        step_all(self.optimizers.values(), set_to_none=True, visibility=vis)
        # ExponentialLR on the means
        self.optimizers["means"].param_groups[0]["lr"] = self.means_lr0 * self.lr_gamma ** (step + 1)
        # densification
        s = cfg.strategy
        if isinstance(s, DefaultStrategy):
            refine_now = (step < s.refine_stop_iter and step > s.refine_start_iter and step % s.refine_every == 0
                          and step % s.reset_every >= s.pause_refine_after_reset)
            if refine_now and self.world_size > 1:
                # local statistics of this step are added inside step_post_backward; reduce after it
                s._update_state(self.splats, self.strategy_state, info, packed=cfg.packed)
                sdist.all_reduce_strategy_state(self.strategy_state)
                self._post_backward_refine_only(step, info)
            else:
                s.step_post_backward(params=self.splats, optimizers=self.optimizers, state=self.strategy_state,
                                     step=step, info=info, packed=cfg.packed, generator=self._split_gen)
        elif isinstance(s, MCMCStrategy):
            s.step_post_backward(params=self.splats, optimizers=self.optimizers, state=self.strategy_state, step=step,
                                 info=info, lr=self.optimizers["means"].param_groups[0]["lr"],
                                 generator=self._split_gen)
        self.last_info = info
        self.step += 1
        return loss.detach()

    def _post_backward_refine_only(self, step: int, info: dict) -> None:
        s = self.cfg.strategy
        n_dupli, n_split = s._grow_gs(self.splats, self.optimizers, self.strategy_state, step, self._split_gen)
        n_prune = s._prune_gs(self.splats, self.optimizers, self.strategy_state, step)
        if s.verbose:
            print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split, {n_prune} GSs pruned. "
                  f"Now having {len(self.splats['means'])} GSs.")
        self.strategy_state["grad2d"].zero_()
        self.strategy_state["count"].zero_()
        if step % s.reset_every == 0 and step > 0:
            from .strategy import reset_opa
            reset_opa(params=self.splats, optimizers=self.optimizers, state=self.strategy_state,
                      value=s.prune_opa * 2.0)

    def train(self, max_steps: Optional[int] = None) -> None:
        """Loop over `views` (batch_size per step, rank-strided like a DistributedSampler)."""
        assert self.views, "Runner.train needs views"
        cfg = self.cfg
        n = max_steps if max_steps is not None else cfg.max_steps
        B = cfg.batch_size
        dev = self.device
        order = list(range(self.world_rank, len(self.views), self.world_size)) or [0]
        cursor = 0
        for _ in range(n):
            if self.stop_training:
                break
            batch = [self.views[order[(cursor + i) % len(order)]] for i in range(B)]
            cursor += B
            c2w = torch.stack([b["camtoworld"] for b in batch]).to(dev)
            Ks = torch.stack([b["K"] for b in batch]).to(dev)
            pixels = torch.stack([b["image"] for b in batch]).to(dev) / 255.0
            self.train_step(c2w, Ks, pixels)
